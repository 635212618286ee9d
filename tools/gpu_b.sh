#!/bin/bash
# EXPERIMENT: row-complete k = 1 GEMM with the weight stage through registers (FS2_BV=1, 128-row tiles) against LDS-DMA for both operands
mkdir -p gpurun_out/r3v
python -m pytest tests/test_gpu_ops.py -x -q -k "row_complete" 2>&1 | tail -2
FS2_BV=1 python -m pytest tests/test_gpu_ops.py -x -q -k "row_complete" 2>&1 | tail -2
for wl in c3 c4; do for bv in 0 1; do
  FS2_MT8=2 FS2_BV=$bv python bench.py --no-cpu-baseline --workload $wl --profile-kernels > /dev/null 2> gpurun_out/r3v/sites_${wl}_bv${bv}.txt
  FS2_MT8=2 FS2_BV=$bv python bench.py --no-cpu-baseline --workload $wl > gpurun_out/r3v/bench_${wl}_bv${bv}.json 2>/dev/null
done; done
for f in gpurun_out/r3v/bench_*.json; do echo "$f $(python -c "import json,sys;d=json.load(open('$f'));print(d['value'], d['ms_per_step'], d.get('mel_max_abs_diff'))")"; done
for f in gpurun_out/r3v/sites_*.txt; do echo $f; grep -E "dec.ffn2_ln|dec.out_ln|dec.in |enc.ffn2_ln|enc.out_ln" $f | cut -c1-90; done
