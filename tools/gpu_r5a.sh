#!/bin/bash
# round 5, GPU session A: gemm_row4_bf16 -- stand-alone probe (bit-identity vs gemm_row8_bf16, times, phase stamps), in-model A/B, a first test pass
D=gpurun_out/r5a
mkdir -p $D
P=tools/probes
for spec in "36611 384 20" "36611 1024 20" "78211 1024 10" "470011 384 4" "470011 1024 4"; do set -- $spec
  timeout 120 $P/row_probe.bin $1 $2 $3 > $D/probe_$1_$2.txt 2>&1
  echo "== probe $1 $2 rc=$?"; grep -E "bit-identity|EPI [012]  row" $D/probe_$1_$2.txt | head -30
done
for spec in "36611 384 5" "36611 1024 5"; do set -- $spec
  timeout 120 $P/row_probe_timing.bin $1 $2 $3 > $D/probe_timing_$1_$2.txt 2>&1
  grep -A12 "phase stamps" $D/probe_timing_$1_$2.txt
done
for r4 in 0 1; do
  FS2_ROW4=$r4 timeout 300 python bench.py --no-cpu-baseline --profile-kernels > $D/bench_c3_row4_$r4.json 2> $D/sites_c3_row4_$r4.txt
  FS2_ROW4=$r4 timeout 300 python bench.py --no-cpu-baseline > $D/bench_c3_plain_row4_$r4.json 2>/dev/null
done
for r4 in 0 1; do
  FS2_ROW4=$r4 timeout 300 python bench.py --no-cpu-baseline --workload c4 --profile-kernels > $D/bench_c4_row4_$r4.json 2> $D/sites_c4_row4_$r4.txt
done
FS2_ROW4=1 FS2_SCHED4=0 timeout 300 python bench.py --no-cpu-baseline --profile-kernels > $D/bench_c3_row4_s0.json 2> $D/sites_c3_row4_s0.txt
FS2_ROW4=1 FS2_MT4=4 timeout 300 python bench.py --no-cpu-baseline --profile-kernels > $D/bench_c3_row4_mt4.json 2> $D/sites_c3_row4_mt4.txt
for f in $D/sites_*.txt; do echo "== $f"; grep -E "dec.ffn2_ln|dec.out_ln|dec.in " $f; done
for f in $D/bench_*.json; do echo "$f $(python -c "import json,sys;d=json.load(open('$f'));print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)"; done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "g1 or g2 or c2 or c3 or device_driven or tile_heights" 2>&1 | tail -8 > $D/pytest_subset.txt
tail -n 5 $D/pytest_subset.txt
