#!/bin/bash
# round 3, GPU call A: full -m gpu suite + c3 in bf16x3 / mix_mx (per-site tables)
mkdir -p gpurun_out/r3a
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r3a/pytest.txt
for p in bf16x3 mix_mx; do
  python bench.py --precision $p --no-cpu-baseline > gpurun_out/r3a/bench_c3_$p.json 2> gpurun_out/r3a/bench_c3_$p.err
  python bench.py --precision $p --no-cpu-baseline --profile-kernels > gpurun_out/r3a/sites_c3_$p.json 2> gpurun_out/r3a/sites_c3_$p.txt
done
python bench.py --precision mix_mx > gpurun_out/r3a/bench_c3_mix_mx_full.json 2> gpurun_out/r3a/bench_c3_mix_mx_full.err
tail -5 gpurun_out/r3a/pytest.txt
cat gpurun_out/r3a/bench_c3_bf16x3.json gpurun_out/r3a/bench_c3_mix_mx.json | cut -c1-400
