#!/bin/bash
mkdir -p gpurun_out/r3h
python -m pytest tests/test_gpu_parity.py -x -q -k "c2 or c3_free or g1 or device_driven or c1_single" 2>&1 | tail -4 > gpurun_out/r3h/pytest_par.txt
tail -n 4 gpurun_out/r3h/pytest_par.txt
for fv in 1 0 1 0; do
FS2_FUSE_VAR=$fv python bench.py --no-cpu-baseline > gpurun_out/r3h/bench_c3_fuse$fv.json 2>/dev/null
FS2_FUSE_VAR=$fv python bench.py --no-cpu-baseline --workload c1 > gpurun_out/r3h/bench_c1_fuse$fv.json 2>/dev/null
echo "fuse=$fv c3 $(python -c "import json;d=json.load(open('gpurun_out/r3h/bench_c3_fuse$fv.json'));print(d['value'], d['ms_per_step'])") c1 $(python -c "import json;d=json.load(open('gpurun_out/r3h/bench_c1_fuse$fv.json'));print(d['value'], d['ms_per_step'])")"
done
python bench.py --no-cpu-baseline --profile-kernels 2>&1 >/dev/null | grep "var\.\|energy\|pitch"
