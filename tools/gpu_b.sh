#!/bin/bash
# attention with software-pipelined fragment reads: correctness + site times
mkdir -p gpurun_out/r3t
python -m pytest tests/test_gpu_ops.py -x -q -k "attention" 2>&1 | tail -3 > gpurun_out/r3t/pytest_ops.txt
python -m pytest tests/test_gpu_parity.py -x -q -k "c2_batch or g1_ or g3_ or batch_invariance or config_variants or c1_single" 2>&1 | tail -3 > gpurun_out/r3t/pytest_par.txt
for wl in c3 c4 c1; do
  python bench.py --no-cpu-baseline --workload $wl --profile-kernels > gpurun_out/r3t/sites_${wl}.json 2> gpurun_out/r3t/sites_${wl}.txt
  python bench.py --no-cpu-baseline --workload $wl > gpurun_out/r3t/bench_${wl}.json 2>/dev/null
done
cat gpurun_out/r3t/pytest_ops.txt gpurun_out/r3t/pytest_par.txt
for f in gpurun_out/r3t/bench_*.json; do echo "$f $(python -c "import json,sys;d=json.load(open('$f'));print(d['value'], d['ms_per_step'])")"; done
for f in gpurun_out/r3t/sites_*.txt; do echo $f; grep -E "dec.attn|enc.attn" $f | cut -c1-90; done
