#!/bin/bash
mkdir -p gpurun_out/r3e
for wl in c4 c3; do for mt in 2 -1; do
FS2_MT8=$mt python bench.py --no-cpu-baseline --workload $wl > gpurun_out/r3e/ab_${wl}_mt$mt.json 2>/dev/null
echo "$wl FS2_MT8=$mt: $(python -c "import json;d=json.load(open('gpurun_out/r3e/ab_${wl}_mt$mt.json'));print(d['value'], d['ms_per_step'])")"
done; done
FS2_MT8=2 python bench.py --no-cpu-baseline --workload c4 --profile-kernels 2>&1 >/dev/null | grep -v amdgpu | head -6
