#!/bin/bash
# round 5, GPU session O: the default bench line once more (with the timed-output identity field), three times for the spread
D=gpurun_out/r5o
mkdir -p $D
for i in 1 2 3; do
  python bench.py > $D/bench_c3_$i.json 2>$D/bench_c3_$i.err
  python -c "import json;d=json.load(open('$D/bench_c3_$i.json'));print(d['value'],d['ms_per_step'],d['sustained_ms_per_step'],d.get('sclk_mhz_median'),d['one_stream']['value'],d['roofline']['frac'],d['mel_max_abs_diff'],d.get('timed_output_identical_to_checked'),d['vs_cpu'])"
done
