#!/bin/bash
mkdir -p gpurun_out/r3i
python bench.py --no-cpu-baseline > gpurun_out/r3i/bench_c3.json 2>/dev/null
python bench.py --no-cpu-baseline --workload c5 > gpurun_out/r3i/bench_c5.json 2>/dev/null
python bench.py --no-cpu-baseline --workload c2 > gpurun_out/r3i/bench_c2.json 2>/dev/null
for f in gpurun_out/r3i/bench_*.json; do echo "$f $(python -c "import json,sys;d=json.load(open('$f'));print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])")"; done
python -m pytest tests/test_gpu_parity.py -x -q -k "device_driven or sharded or c5" 2>&1 | tail -2
