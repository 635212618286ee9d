#!/bin/bash
# row-complete k = 1 GEMM: A tiles requested two steps ahead (ring of three, FS2_A_RING=3) against one step (2)
mkdir -p gpurun_out/r3w
python -m pytest tests/test_gpu_ops.py -x -q -k "row_complete" 2>&1 | tail -2
python -m pytest tests/test_gpu_parity.py -x -q -k "c2_batch or c2_row_complete" 2>&1 | tail -2
for wl in c3 c4; do for ar in 2 3 2 3; do
  n=$(ls gpurun_out/r3w | grep -c "bench_${wl}_ar${ar}")
  FS2_MT8=2 FS2_A_RING=$ar python bench.py --no-cpu-baseline --workload $wl > gpurun_out/r3w/bench_${wl}_ar${ar}_$n.json 2>/dev/null
done; done
for wl in c3 c4; do for ar in 2 3; do
  FS2_MT8=2 FS2_A_RING=$ar python bench.py --no-cpu-baseline --workload $wl --profile-kernels > /dev/null 2> gpurun_out/r3w/sites_${wl}_ar${ar}.txt
done; done
python bench.py --no-cpu-baseline --workload c3 > gpurun_out/r3w/bench_c3_auto.json 2>/dev/null
for f in gpurun_out/r3w/bench_*.json; do echo "$f $(python -c "import json,sys;d=json.load(open('$f'));print(d['value'], d['ms_per_step'])")"; done
for f in gpurun_out/r3w/sites_*.txt; do echo $f; grep -E "dec.ffn2_ln|dec.out_ln|dec.in " $f | cut -c1-90; done
