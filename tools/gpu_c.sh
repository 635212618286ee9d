#!/bin/bash
# round 3, GPU call C: full -m gpu suite + c3/c1/c4 benches in the new default
mkdir -p gpurun_out/r3d
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3d/pytest.txt
python bench.py > gpurun_out/r3d/bench_c3_default.json 2> gpurun_out/r3d/bench_c3_default.err
python bench.py --no-cpu-baseline --profile-kernels > gpurun_out/r3d/sites_c3_mix_mx.json 2> gpurun_out/r3d/sites_c3_mix_mx.txt
python bench.py --precision bf16x3 --no-cpu-baseline > gpurun_out/r3d/bench_c3_bf16x3.json 2>/dev/null
python bench.py --precision bf16x3 --no-cpu-baseline --profile-kernels > /dev/null 2> gpurun_out/r3d/sites_c3_bf16x3.txt
python bench.py --workload c1 --no-cpu-baseline > gpurun_out/r3d/bench_c1.json 2>/dev/null
python bench.py --workload c4 --no-cpu-baseline > gpurun_out/r3d/bench_c4.json 2>/dev/null
tail -4 gpurun_out/r3d/pytest.txt
for f in gpurun_out/r3d/bench_*.json; do echo $f; cut -c1-190 $f; done
grep -v amdgpu gpurun_out/r3d/sites_c3_mix_mx.txt | head -12
