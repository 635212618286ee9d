#!/bin/bash
# round 5, GPU session L: bench defaults with several steps in flight (3 streams on one GPU, 2 per rank with the collective)
D=gpurun_out/r5l
mkdir -p $D
for wl in c3 c2 c5 c4 c1; do
  timeout 300 python bench.py --no-cpu-baseline --workload $wl > $D/bench_${wl}.json 2>$D/bench_${wl}.err
done
FS2_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --workload c5 > $D/bench_c5_rccl1.json 2>$D/bench_c5_rccl1.err
FS2_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --workload c5 --streams 1 > $D/bench_c5_rccl1_s1.json 2>/dev/null
FS2_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --workload c5 --streams 3 > $D/bench_c5_rccl1_s3.json 2>/dev/null
FS2_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --workload c5 --padded > $D/bench_c5_rccl1_padded.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --streams 1 > $D/bench_c3_s1.json 2>/dev/null
for f in $D/bench_*.json; do echo "$f $(python -c "import json,sys;d=json.load(open('$f'));print(d['value'], d['ms_per_step'], d['ms_per_step_median'], d.get('sustained_ms_per_step'), d['config'].get('overlap_encoder'), d['config'].get('streams'), (d.get('one_stream') or {}).get('value'), d['roofline']['kernel'], d['roofline']['frac'], (d['roofline'].get('one_step_in_flight') or {}).get('frac'))" 2>&1 | tail -1)"; done
tail -3 $D/bench_c3.err $D/bench_c5_rccl1.err
