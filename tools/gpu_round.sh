#!/bin/bash
# The GPU stages of a round, each sized for one gpurun call (the ONE maintained session script: rounds 1-5 left 20 one-off copies behind).
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_round.sh <tag> <stage>...'        tag: r06
# Stages:
#   tests    full `pytest -m gpu` + smoke(); measured errors -> <out>/<tag>_measured_errors.jsonl
#   bench    bench lines of every config and arithmetic mode + launch-site tables + schedule A/Bs + c1 latency
#   profile  rocprofv3 kernel trace + PMC passes (c3 default, c4), c1 trace.  Run LAST on the final kernel sources: profiles/<tag>_traffic.json
#            carries their hash and bench.py quotes the record only while it matches.
#   probes   the stand-alone kernel probes (row kernels, attention incl. the P.V MFMA-count ablation, conv incl. the fp4 timing probe, fp4 semantics);
#            binaries are built on the build host first:  bash tools/gpu_round.sh <tag> build-probes
# Output: gpurun_out/<tag>/ (copy what is to be judged into profiles/).
TAG=${1:-r06}; shift
D=gpurun_out/$TAG
mkdir -p $D
P=tools/probes
for stage in "$@"; do
case $stage in
build-probes)
  HF="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -I fastspeech2_amd/csrc -I $P"
  python $P/make_fp6_probe.py && python $P/make_fp4_probe.py
  /opt/rocm/bin/hipcc $HF $P/row_probe.hip -o $P/row_probe.bin &
  /opt/rocm/bin/hipcc $HF $P/mx_conv_probe.hip -o $P/mx_conv_probe.bin &
  /opt/rocm/bin/hipcc $HF $P/fp4_probe.hip -o $P/fp4_probe.bin &
  /opt/rocm/bin/hipcc $HF -DFS2_W32_ABL=0 $P/attn_w32_probe.hip -o $P/attn_w32_probe_abl0.bin &
  /opt/rocm/bin/hipcc $HF -DFS2_W32_ABL=256 $P/attn_w32_probe.hip -o $P/attn_w32_probe_abl256.bin &
  /opt/rocm/bin/hipcc $HF -DFS2_CONV4_TIMING $P/conv4_probe.hip -o $P/conv4_probe.bin &      # the rejected one-wave-per-SIMD conv (profiles/r06_conv4_probe.txt)
  wait; ls -la $P/*.bin
  ;;
tests)
  rm -f gpurun_out/measured_errors.jsonl
  python -m pytest tests -m gpu -x -q 2>&1 | grep -v Warning | tail -8 > $D/${TAG}_pytest_gpu_summary.txt
  cp gpurun_out/measured_errors.jsonl $D/${TAG}_measured_errors.jsonl 2>/dev/null
  python -c "import __graft_entry__ as g; g.smoke()" > $D/smoke.txt 2>&1
  tail -n 3 $D/${TAG}_pytest_gpu_summary.txt; tail -n 2 $D/smoke.txt
  ;;
bench)
  python bench.py > $D/bench_${TAG}_c3_mix_mx4.json 2> $D/bench_c3.err
  for spec in "c3 mix_mx" "c3 bf16x3" "c3 fp32" "c2 mix_mx4" "c2 fp32" "c1 mix_mx4" "c4 mix_mx4" "c4 mix_mx" "c4 bf16x3" "c5 mix_mx4" "c5 mix_mx"; do set -- $spec
    python bench.py --workload $1 --precision $2 --no-cpu-baseline > $D/bench_${TAG}_$1_$2.json 2>/dev/null
  done
  for wl in c3 c2 c4 c5; do      # the schedule's A/B: one stream + overlap_encoder; and c3 with every launch of a step on one stream (rounds 1-4's schedule as the TIMED region)
    python bench.py --workload $wl --no-cpu-baseline --streams 1 > $D/bench_${TAG}_${wl}_mix_mx4_one_stream.json 2>/dev/null
  done
  python bench.py --no-cpu-baseline --streams 1 --no-overlap-encoder > $D/bench_${TAG}_c3_mix_mx4_one_stream_no_overlap.json 2>/dev/null
  python bench.py --workload c1 --no-cpu-baseline --graph > $D/bench_${TAG}_c1_mix_mx4_graph.json 2>/dev/null
  FS2_FORCE_DIST=1 python bench.py --workload c5 --no-cpu-baseline > $D/bench_${TAG}_c5_rccl_single_rank.json 2>/dev/null
  python bench.py --workload c5 --no-cpu-baseline --regime-utterances 1024 > $D/bench_${TAG}_c5_mix_mx4_regime_of_1024.json 2>/dev/null      # the shard with the whole batch's kernel variants (what every rank runs)
  FS2_ROW4=0 FS2_FFN2_MX=0 python bench.py --precision mix_mx --no-cpu-baseline > $D/bench_${TAG}_c3_mix_mx_row8_fp32_residual.json 2>/dev/null   # rounds 1-4's row kernels, fp32 residual rows
  for wl in c3 c4 c2 c1; do
    python bench.py --no-cpu-baseline --sustain 0 --workload $wl --profile-kernels > /dev/null 2> $D/${TAG}_${wl}_launch_sites_hipevents.txt
  done
  python bench.py --no-cpu-baseline --sustain 0 --precision mix_mx --profile-kernels > /dev/null 2> $D/${TAG}_c3_launch_sites_hipevents_mix_mx.txt
  python tools/latency_c1.py > $D/${TAG}_latency_c1.txt 2>&1; python tools/launch_floor.py 85 >> $D/${TAG}_latency_c1.txt 2>&1
  for f in $D/bench_${TAG}_*.json; do echo "$f $(python -c "import json,sys;d=json.load(open('$f'));print(d['value'], d['ms_per_step'], d.get('value_one_forward'), d.get('ms_per_forward'), d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])")"; done
  ;;
profile)
  bash tools/profile_round.sh ${TAG} c3 mix_mx4 1 > $D/profile_c3.log 2>&1
  bash tools/profile_round.sh ${TAG}c4 c4 mix_mx4 1 > $D/profile_c4.log 2>&1
  bash tools/profile_round.sh ${TAG}c1 c1 mix_mx4 0 > $D/profile_c1.log 2>&1
  bash tools/profile_round.sh ${TAG}one c3 mix_mx4 0 "--streams 1 --no-overlap-encoder --steps 20 --warmup 5" > $D/profile_c3_one_stream.log 2>&1
  ls gpurun_out/prof_${TAG} gpurun_out/prof_${TAG}c4 gpurun_out/prof_${TAG}c1 gpurun_out/prof_${TAG}one
  ;;
probes)
  $P/fp4_probe.bin > $D/${TAG}_fp4_probe.txt 2>&1
  for k in 384 1024; do $P/row_probe.bin 36611 $k 30 > $D/${TAG}_row_probe_c3_K$k.txt 2>&1; done
  $P/row_probe.bin 456700 1024 5 > $D/${TAG}_row_probe_c4_K1024.txt 2>&1
  for r in 1 2; do for abl in 0 256; do
    echo "FS2_W32_ABL=$abl c4-like (B=256, L 250..4030):"; $P/attn_w32_probe_abl$abl.bin 256 250 4030 192 5 | tail -1
    echo "FS2_W32_ABL=$abl c3-like (B=64, L 300..900):"; $P/attn_w32_probe_abl$abl.bin 64 300 900 192 20 | tail -1
  done; done > $D/${TAG}_attn_pv_ablation.txt 2>&1
  for r in 1 2; do for bm in 256 4256 2256; do $P/mx_conv_probe.bin 36611 384 1024 9 $bm 2 0 | tail -1; done; $P/mx_conv_probe.bin 36611 384 1024 9 256 0 0 | tail -1; done > $D/${TAG}_fp4_conv_probe.txt 2>&1
  for r in 1 2; do for bm in 256 4256; do $P/mx_conv_probe.bin 456700 384 1024 9 $bm 2 0 | tail -1; done; done >> $D/${TAG}_fp4_conv_probe.txt 2>&1
  tail -3 $D/${TAG}_fp4_conv_probe.txt
  for rows in 35636 307000 76600; do $P/conv4_probe.bin $rows; done > $D/${TAG}_conv4_probe_raw.txt 2>&1      # c3 / c4 / c5-shard row counts
  ;;
esac
done
