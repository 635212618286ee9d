#!/bin/bash
# round 5 GPU sessions, by stage (each fits one gpurun call):
#   tests    full -m gpu suite + smoke
#   bench    benches of every config (+ launch-site tables, c1 latency)
#   profile  rocprofv3 kernel trace + PMC passes (c3 default, c4), c1 trace -- run LAST: profiles/r05_traffic.json is tied to the kernel sources' hash
D=gpurun_out/r5z
mkdir -p $D
for stage in "$@"; do
case $stage in
tests)
  rm -f gpurun_out/measured_errors.jsonl
  python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $D/pytest.txt
  cp gpurun_out/measured_errors.jsonl $D/r05_measured_errors.jsonl 2>/dev/null
  python -c "import __graft_entry__ as g; g.smoke()" > $D/smoke.txt 2>&1
  tail -n 3 $D/pytest.txt; tail -n 2 $D/smoke.txt
  ;;
bench)
  python bench.py > $D/bench_r05_c3_mix_mx.json 2> $D/bench_c3.err
  for spec in "c3 bf16x3" "c3 fp32" "c2 mix_mx" "c2 fp32" "c1 mix_mx" "c4 mix_mx" "c4 bf16x3" "c5 mix_mx"; do set -- $spec
    python bench.py --workload $1 --precision $2 --no-cpu-baseline > $D/bench_r05_$1_$2.json 2>/dev/null
  done
  for wl in c3 c2 c4 c5; do      # the schedule's A/B: one stream (+ overlap_encoder), and c3 with every launch of a step on one stream (rounds 1-4)
    python bench.py --workload $wl --no-cpu-baseline --streams 1 > $D/bench_r05_${wl}_mix_mx_one_stream.json 2>/dev/null
  done
  python bench.py --no-cpu-baseline --streams 1 --no-overlap-encoder > $D/bench_r05_c3_mix_mx_one_stream_no_overlap.json 2>/dev/null
  python bench.py --workload c1 --no-cpu-baseline --graph > $D/bench_r05_c1_mix_mx_graph.json 2>/dev/null
  FS2_FORCE_DIST=1 python bench.py --workload c5 --no-cpu-baseline > $D/bench_r05_c5_rccl_single_rank.json 2>/dev/null
  FS2_ROW4=0 FS2_FFN2_MX=0 python bench.py --no-cpu-baseline > $D/bench_r05_c3_mix_mx_row8.json 2>/dev/null
  FS2_ROW4=0 FS2_FFN2_MX=0 python bench.py --workload c4 --no-cpu-baseline > $D/bench_r05_c4_mix_mx_row8.json 2>/dev/null
  python bench.py --no-cpu-baseline --sustain 0 --profile-kernels > /dev/null 2> $D/r05_launch_sites_hipevents.txt
  python bench.py --no-cpu-baseline --sustain 0 --workload c4 --profile-kernels > /dev/null 2> $D/r05_c4_launch_sites_hipevents.txt
  python bench.py --no-cpu-baseline --sustain 0 --workload c2 --profile-kernels > /dev/null 2> $D/r05_c2_launch_sites_hipevents.txt
  python bench.py --no-cpu-baseline --sustain 0 --workload c1 --profile-kernels > /dev/null 2> $D/r05_c1_launch_sites_hipevents.txt
  FS2_ROW8=1 python bench.py --workload c2 --no-cpu-baseline > $D/bench_r05_c2_mix_mx_rowkernels_forced.json 2>/dev/null
  FS2_ROW8=1 FS2_QKV8=1 python bench.py --workload c2 --no-cpu-baseline > $D/bench_r05_c2_mix_mx_rowkernels_qkv8_forced.json 2>/dev/null
  FS2_ROW8=1 python bench.py --no-cpu-baseline --sustain 0 --workload c2 --profile-kernels > /dev/null 2> $D/r05_c2_launch_sites_hipevents_rowkernels_forced.txt
  python tools/latency_c1.py > $D/latency_c1.txt 2>&1
  for f in $D/bench_r05_*.json; do echo "$f $(python -c "import json,sys;d=json.load(open('$f'));print(d['value'], d['ms_per_step'], d.get('sustained_ms_per_step'), d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])")"; done
  ;;
profile)
  bash tools/profile_round.sh r05 c3 mix_mx 1 > $D/profile_c3.log 2>&1
  bash tools/profile_round.sh r05c4 c4 mix_mx 1 > $D/profile_c4.log 2>&1
  bash tools/profile_round.sh r05c1 c1 mix_mx 0 > $D/profile_c1.log 2>&1
  ls gpurun_out/prof_r05 gpurun_out/prof_r05c4 gpurun_out/prof_r05c1
  ;;
esac
done
