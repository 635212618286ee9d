#!/bin/bash
# round 5, GPU session J: two whole steps in flight (steps on alternating streams) -- test, schedule matrix, bench A/B
D=gpurun_out/r5j
mkdir -p $D
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_steps_in_flight or overlap_encoder" 2>&1 | tail -6 > $D/pytest.txt
tail -n 5 $D/pytest.txt
timeout 600 python tools/debug/two_decode_streams.py > $D/matrix.txt 2>&1
cat $D/matrix.txt
for wl in c3 c2; do
  timeout 300 python bench.py --no-cpu-baseline --workload $wl > $D/bench_${wl}_2s_ov.json 2>$D/bench_${wl}_2s_ov.err
  timeout 300 python bench.py --no-cpu-baseline --workload $wl --no-overlap-encoder > $D/bench_${wl}_2s_noov.json 2>/dev/null
done
for f in $D/bench_*.json; do echo "$f $(python -c "import json,sys;d=json.load(open('$f'));print(d['value'], d['ms_per_step'], d['ms_per_step_median'], d.get('sustained_ms_per_step'), d['config'].get('overlap_encoder'), d['config'].get('streams'), d.get('one_stream'), d['roofline']['frac'], d['roofline'].get('one_step_in_flight'))" 2>&1 | tail -1)"; done
tail -3 $D/bench_c3_2s_ov.err
