#!/bin/bash
mkdir -p gpurun_out/r3c
for p in mix_mx bf16x3; do
  python bench.py --precision $p --no-cpu-baseline > gpurun_out/r3c/bench_c3_$p.json 2> gpurun_out/r3c/bench_c3_$p.err
  python bench.py --precision $p --no-cpu-baseline --profile-kernels > gpurun_out/r3c/sites_c3_$p.json 2> gpurun_out/r3c/sites_c3_$p.txt
done
python bench.py --precision mix_mx > gpurun_out/r3c/bench_c3_mix_mx_full.json 2> gpurun_out/r3c/bench_c3_mix_mx_full.err
python bench.py --precision mix_mx --workload c4 --no-cpu-baseline > gpurun_out/r3c/bench_c4_mix_mx.json 2>/dev/null
python bench.py --precision bf16x3 --workload c4 --no-cpu-baseline > gpurun_out/r3c/bench_c4_bf16x3.json 2>/dev/null
for f in gpurun_out/r3c/bench_*.json; do echo $f; cut -c1-200 $f; done
grep -v amdgpu gpurun_out/r3c/sites_c3_mix_mx.txt | head -8
