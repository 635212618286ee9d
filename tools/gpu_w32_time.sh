#!/bin/bash
# attn_w32 timing: c3- / c4- / c2-like against attn_bf16, then the phase breakdown of wave 0 (instrumented build)
mkdir -p gpurun_out/w32
P=tools/probes/attn_w32_probe.bin
T=tools/probes/attn_w32_probe_timing.bin
{
echo "--- correctness spot checks"; timeout 60 $P 9 100 333 192 1 5; timeout 60 $P 4 100 300 192 1 0 12
echo "--- c3-like"; timeout 60 $P 64 300 800 192 5 0
echo "--- c2-like"; timeout 60 $P 16 500 900 192 5 0
echo "--- c4-like"; timeout 120 $P 256 250 3300 192 3 0
echo "--- one utterance"; timeout 60 $P 1 650 650 192 5 0
echo "--- phases, c3-like"; timeout 60 $T 64 300 800 192 3 0
echo "--- phases, c4-like"; timeout 120 $T 256 250 3300 192 2 0
echo "--- phases, one workgroup per CU exactly (256 x 128 queries x 1 head pair)"; timeout 60 $T 128 128 128 192 3 0
} > gpurun_out/w32/time_$1.txt 2>&1
cat gpurun_out/w32/time_$1.txt
