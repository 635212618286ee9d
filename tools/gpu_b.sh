#!/bin/bash
mkdir -p gpurun_out/r3f
python -m pytest tests/test_gpu_parity.py -x -q -k "config_variants or g5 or g7 or checkpoint" 2>&1 | tail -12 > gpurun_out/r3f/pytest_a.txt
tail -n 12 gpurun_out/r3f/pytest_a.txt
