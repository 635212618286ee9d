#!/bin/bash
mkdir -p gpurun_out/r3g
python -m pytest tests/test_gpu_ops.py -x -q -k "attention" 2>&1 | tail -3 > gpurun_out/r3g/pytest_attn.txt
python -m pytest tests/test_gpu_parity.py -x -q -k "c2_batch or c1_single or g9 or c4" 2>&1 | tail -3 > gpurun_out/r3g/pytest_par.txt
python bench.py --no-cpu-baseline --profile-kernels > /dev/null 2> gpurun_out/r3g/sites_c3.txt
python bench.py --no-cpu-baseline --workload c4 --profile-kernels > /dev/null 2> gpurun_out/r3g/sites_c4.txt
python bench.py --no-cpu-baseline --workload c4 > gpurun_out/r3g/bench_c4.json 2>/dev/null
python bench.py --no-cpu-baseline --workload c1 > gpurun_out/r3g/bench_c1.json 2>/dev/null
python bench.py --no-cpu-baseline > gpurun_out/r3g/bench_c3.json 2>/dev/null
tail -n 2 gpurun_out/r3g/pytest_attn.txt gpurun_out/r3g/pytest_par.txt
for f in gpurun_out/r3g/bench_*.json; do cut -c1-170 $f; done
grep -v amdgpu gpurun_out/r3g/sites_c3.txt | head -6
grep -v amdgpu gpurun_out/r3g/sites_c4.txt | head -6
