#!/bin/bash
# round 3 final GPU session: full -m gpu suite, smoke, benches of every config, rocprofv3 kernel trace + PMC passes (c3 default, c4), c1 trace
mkdir -p gpurun_out/r3z
rm -f gpurun_out/measured_errors.jsonl
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r3z/pytest.txt
cp gpurun_out/measured_errors.jsonl gpurun_out/r3z/r03_measured_errors.jsonl 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3z/smoke.txt 2>&1
python bench.py > gpurun_out/r3z/bench_r03_c3_mix_mx.json 2> gpurun_out/r3z/bench_c3.err
for spec in "c3 bf16x3" "c3 fp32" "c2 mix_mx" "c2 fp32" "c1 mix_mx" "c4 mix_mx" "c4 bf16x3" "c5 mix_mx"; do set -- $spec
  python bench.py --workload $1 --precision $2 --no-cpu-baseline > gpurun_out/r3z/bench_r03_$1_$2.json 2>/dev/null
done
python bench.py --workload c1 --no-cpu-baseline --graph > gpurun_out/r3z/bench_r03_c1_mix_mx_graph.json 2>/dev/null
FS2_FORCE_DIST=1 python bench.py --workload c5 --no-cpu-baseline > gpurun_out/r3z/bench_r03_c5_rccl_single_rank.json 2>/dev/null
python bench.py --no-cpu-baseline --profile-kernels > /dev/null 2> gpurun_out/r3z/r03_launch_sites_hipevents.txt
python bench.py --no-cpu-baseline --workload c4 --profile-kernels > /dev/null 2> gpurun_out/r3z/r03_c4_launch_sites_hipevents.txt
python bench.py --no-cpu-baseline --workload c1 --profile-kernels > /dev/null 2> gpurun_out/r3z/r03_c1_launch_sites_hipevents.txt
python tools/latency_c1.py > gpurun_out/r3z/latency_c1.txt 2>&1
bash tools/profile_round.sh r03 c3 mix_mx 1 > gpurun_out/r3z/profile_c3.log 2>&1
bash tools/profile_round.sh r03c4 c4 mix_mx 1 > gpurun_out/r3z/profile_c4.log 2>&1
bash tools/profile_round.sh r03c1 c1 mix_mx 0 > gpurun_out/r3z/profile_c1.log 2>&1
tail -n 3 gpurun_out/r3z/pytest.txt; tail -n 2 gpurun_out/r3z/smoke.txt
for f in gpurun_out/r3z/bench_r03_*.json; do echo "$f $(python -c "import json,sys;d=json.load(open('$f'));print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])")"; done
