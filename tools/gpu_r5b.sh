#!/bin/bash
# round 5, GPU session B: gemm_row4_bf16 with scalar-base LDS-DMA pieces and three spreading schedules; the new tests (all of c5, id range, sharded ok())
D=gpurun_out/r5b
mkdir -p $D
P=tools/probes
for spec in "36611 384 20" "36611 1024 20" "470011 384 4" "470011 1024 4"; do set -- $spec
  timeout 120 $P/row_probe.bin $1 $2 $3 > $D/probe_$1_$2.txt 2>&1
  echo "== probe $1 $2 rc=$?"; grep -E "bit-identity|EPI [012]  row" $D/probe_$1_$2.txt | head -30
done
for spec in "36611 384 5" "36611 1024 5"; do set -- $spec
  timeout 120 $P/row_probe_timing.bin $1 $2 $3 > $D/probe_timing_$1_$2.txt 2>&1
  grep -A9 "phase stamps row4<160" $D/probe_timing_$1_$2.txt
done
for tag in "r0:FS2_ROW4=0" "s2:FS2_SCHED4=2" "s0:FS2_SCHED4=0" "s4:FS2_SCHED4=4"; do
  n=${tag%%:*}; e=${tag#*:}
  env $e timeout 300 python bench.py --no-cpu-baseline --sustain 0 --profile-kernels > $D/bench_c3_sites_$n.json 2> $D/sites_c3_$n.txt
  env $e timeout 300 python bench.py --no-cpu-baseline --sustain 0 > $D/bench_c3_$n.json 2>/dev/null
done
for tag in "r0:FS2_ROW4=0" "s2:FS2_SCHED4=2"; do
  n=${tag%%:*}; e=${tag#*:}
  env $e timeout 300 python bench.py --no-cpu-baseline --sustain 0 --workload c4 --profile-kernels > $D/bench_c4_sites_$n.json 2> $D/sites_c4_$n.txt
  env $e timeout 300 python bench.py --no-cpu-baseline --sustain 0 --workload c4 > $D/bench_c4_$n.json 2>/dev/null
done
for f in $D/sites_*.txt; do echo "== $f"; grep -E "dec.ffn2_ln|dec.out_ln|dec.in " $f; done
for f in $D/bench_c*.json; do echo "$f $(python -c "import json,sys;d=json.load(open('$f'));print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)"; done
timeout 300 python bench.py > $D/bench_default_full.json 2> $D/bench_default_full.err; tail -c 1500 $D/bench_default_full.json
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c5_all or out_of_range or sharded_synthesizer or errors or async_overflow" 2>&1 | tail -15 > $D/pytest_new.txt
tail -n 15 $D/pytest_new.txt
