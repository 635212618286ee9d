#!/bin/bash
# row pass fused into the GEMM (small grids): bit-identity tests, then c1 / c2 / c3 with the switch on and off
mkdir -p gpurun_out/r3u
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "row_pass_fused or c1_single or c2_batch or batch_invariance or device_driven or hip_graph" 2>&1 | tail -5 > gpurun_out/r3u/pytest_a.txt
cat gpurun_out/r3u/pytest_a.txt
for wl in c1 c2 c3; do for f in 1 0; do
  FS2_FUSE_ROWS=$f timeout 300 python bench.py --no-cpu-baseline --workload $wl > gpurun_out/r3u/bench_${wl}_fuse${f}.json 2>/dev/null
done; done
FS2_FUSE_ROWS=1 timeout 300 python bench.py --no-cpu-baseline --workload c1 --graph > gpurun_out/r3u/bench_c1_fuse1_graph.json 2>/dev/null
FS2_FUSE_ROWS=1 timeout 300 python bench.py --no-cpu-baseline --workload c1 --profile-kernels > /dev/null 2> gpurun_out/r3u/sites_c1_fuse1.txt
for f in gpurun_out/r3u/bench_*.json; do echo "$f $(python -c "import json,sys;d=json.load(open('$f'));print(d['value'], d['ms_per_step'])")"; done
head -12 gpurun_out/r3u/sites_c1_fuse1.txt
