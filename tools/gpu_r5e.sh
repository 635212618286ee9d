#!/bin/bash
# round 5, GPU session E: all of c5 (rewritten test)
D=gpurun_out/r5e
mkdir -p $D
rm -f gpurun_out/measured_errors.jsonl
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "c5_all" 2>&1 | tail -25 > $D/pytest.txt
tail -n 14 $D/pytest.txt
cp gpurun_out/measured_errors.jsonl $D/measured_errors.jsonl 2>/dev/null; grep -E "c5_all" $D/measured_errors.jsonl
