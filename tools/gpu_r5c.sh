#!/bin/bash
# round 5, GPU session C: FFN2 + LN2 in the mx arithmetic (gemm_row4_bf16 ARITH = 2): probe, in-model A/B, parity at c3 / c4 / c5
D=gpurun_out/r5c
mkdir -p $D
P=tools/probes
for spec in "36611 1024 20" "36611 384 10"; do set -- $spec
  timeout 160 $P/row_probe.bin $1 $2 $3 > $D/probe_$1_$2.txt 2>&1
  echo "== probe $1 $2 rc=$?"; grep -E "bit-identity|probe:|mx |FFN2-like|EPI 0  row4<160" $D/probe_$1_$2.txt | head -30
done
for tag in "mx0:FS2_FFN2_MX=0" "mx1:FS2_FFN2_MX=1"; do
  n=${tag%%:*}; e=${tag#*:}
  env $e timeout 300 python bench.py --no-cpu-baseline --sustain 0 --profile-kernels > $D/bench_c3_sites_$n.json 2> $D/sites_c3_$n.txt
  env $e timeout 300 python bench.py --sustain 0 > $D/bench_c3_$n.json 2>/dev/null
  env $e timeout 300 python bench.py --no-cpu-baseline --sustain 0 --workload c4 --profile-kernels > $D/bench_c4_sites_$n.json 2> $D/sites_c4_$n.txt
  env $e timeout 300 python bench.py --no-cpu-baseline --sustain 0 --workload c4 > $D/bench_c4_$n.json 2>/dev/null
done
for f in $D/sites_*.txt; do echo "== $f"; grep -E "dec.ffn2_ln|dec.ffn1 |dec.out_ln|dec.in " $f; done
for f in $D/bench_c*.json; do echo "$f $(python -c "import json,sys;d=json.load(open('$f'));print(d['value'], d['ms_per_step'], d.get('mel_max_abs_diff'))" 2>&1 | tail -1)"; done
rm -f gpurun_out/measured_errors.jsonl
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c5_all or full_size_c3 or c3_free or (full_size_c4 and mix_mx) or c2_mixed or device_driven_layout_matches or batch_invariance or heavy_tailed" 2>&1 | tail -15 > $D/pytest_mx.txt
tail -n 15 $D/pytest_mx.txt
cp gpurun_out/measured_errors.jsonl $D/measured_errors.jsonl 2>/dev/null; grep -E "c3_all|c4_all|c5_all" $D/measured_errors.jsonl | tail -8
