#!/bin/bash
# round 5, GPU session I: overlap_encoder (each step's token-level half on a side stream) -- tests and A/B
D=gpurun_out/r5i
mkdir -p $D
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "overlap_encoder or sharded_synthesizer or async_overflow or device_driven_layout_matches" 2>&1 | tail -6 > $D/pytest.txt
tail -n 5 $D/pytest.txt
for wl in c3 c4 c5 c2; do
  timeout 300 python bench.py --no-cpu-baseline --workload $wl --no-overlap-encoder > $D/bench_${wl}_serial.json 2>/dev/null
  timeout 300 python bench.py --no-cpu-baseline --workload $wl > $D/bench_${wl}_overlap.json 2>/dev/null
done
FS2_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --workload c5 --no-overlap-encoder > $D/bench_c5_rccl1_serial.json 2>/dev/null
FS2_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --workload c5 > $D/bench_c5_rccl1_overlap.json 2>/dev/null
timeout 300 python bench.py > $D/bench_c3_default_full.json 2>/dev/null
for f in $D/bench_*.json; do echo "$f $(python -c "import json,sys;d=json.load(open('$f'));print(d['value'], d['ms_per_step'], d.get('sustained_ms_per_step'), d['config'].get('overlap_encoder'), d.get('mel_max_abs_diff'))" 2>&1 | tail -1)"; done
