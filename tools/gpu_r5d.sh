#!/bin/bash
# round 5, GPU session D: tile-relative DMA offsets (the c5 unsharded failure), row4 bit-identity test, all of c5
D=gpurun_out/r5d
mkdir -p $D
true
rm -f gpurun_out/measured_errors.jsonl
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "one_wave_per_simd or c5_all" 2>&1 | tail -15 > $D/pytest.txt
tail -n 12 $D/pytest.txt
cp gpurun_out/measured_errors.jsonl $D/measured_errors.jsonl 2>/dev/null; grep -E "c5_all" $D/measured_errors.jsonl
timeout 300 python bench.py --sustain 0 > $D/bench_c3.json 2>/dev/null; python -c "import json;d=json.load(open('$D/bench_c3.json'));print(d['value'], d['ms_per_step'], d['mel_max_abs_diff'])"
