#!/bin/bash
# round 5, GPU session H: the fused QKV projection's passes on gemm_row4_bf16 (EPI 3): probe (bit-identity vs gemm_qkv8_bf16, times), in-model A/B, tests
D=gpurun_out/r5h
mkdir -p $D
timeout 200 tools/probes/row_probe_qkv.bin 36611 384 20 > $D/probe_qkv_36611.txt 2>&1; echo "probe rc=$?"; grep -E "QKV|qk hi|qk lo|vt hi|vt lo|probe:" $D/probe_qkv_36611.txt
timeout 200 tools/probes/row_probe_qkv.bin 470011 384 4 > $D/probe_qkv_470011.txt 2>&1; echo "probe rc=$?"; grep -E "QKV|probe:" $D/probe_qkv_470011.txt
for tag in "q0:FS2_QKV4=0" "q1:FS2_QKV4=1"; do
  n=${tag%%:*}; e=${tag#*:}
  env $e timeout 300 python bench.py --no-cpu-baseline --sustain 0 --profile-kernels > $D/bench_c3_sites_$n.json 2> $D/sites_c3_$n.txt
  env $e timeout 300 python bench.py --no-cpu-baseline --sustain 0 > $D/bench_c3_$n.json 2>/dev/null
  env $e timeout 300 python bench.py --no-cpu-baseline --sustain 0 --workload c4 --profile-kernels > $D/bench_c4_sites_$n.json 2> $D/sites_c4_$n.txt
  env $e timeout 300 python bench.py --no-cpu-baseline --sustain 0 --workload c4 > $D/bench_c4_$n.json 2>/dev/null
done
for f in $D/sites_*.txt; do echo "== $f"; grep -E "dec.qkv|dec.attn" $f; done
for f in $D/bench_c*.json; do echo "$f $(python -c "import json,sys;d=json.load(open('$f'));print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)"; done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ops.py -m gpu -x -q -k "c2 or full_size_c3 or device_driven or tile_heights or attention or g1 or g3" 2>&1 | tail -6 > $D/pytest.txt
tail -n 5 $D/pytest.txt
