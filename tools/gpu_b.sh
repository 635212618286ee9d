#!/bin/bash
# row-complete kernels after the epilogue rework: correctness + A/B of the tile height (FS2_MT8 2 | 3 | auto) at c3 and c4
mkdir -p gpurun_out/r3r
python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -3 > gpurun_out/r3r/pytest_ops.txt
python -m pytest tests/test_gpu_parity.py -x -q -k "c2_ or g1_ or g3_ or batch_invariance or config_variants" 2>&1 | tail -3 > gpurun_out/r3r/pytest_par.txt
for wl in c3 c4; do for mt in -1 2 3; do
  FS2_MT8=$mt python bench.py --no-cpu-baseline --workload $wl --profile-kernels > gpurun_out/r3r/sites_${wl}_mt${mt}.json 2> gpurun_out/r3r/sites_${wl}_mt${mt}.txt
  FS2_MT8=$mt python bench.py --no-cpu-baseline --workload $wl > gpurun_out/r3r/bench_${wl}_mt${mt}.json 2>/dev/null
done; done
cat gpurun_out/r3r/pytest_ops.txt gpurun_out/r3r/pytest_par.txt
for f in gpurun_out/r3r/bench_*.json; do echo "$f $(python -c "import json,sys;d=json.load(open('$f'));print(d['value'], d['ms_per_step'])")"; done
for f in gpurun_out/r3r/sites_*.txt; do echo $f; grep -E "dec.qkv|dec.ffn2_ln|dec.out_ln|dec.in |var.conv1|pitch.conv0" $f | cut -c1-60; done
