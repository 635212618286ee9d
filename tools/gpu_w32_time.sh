#!/bin/bash
# attn_w32 against attn_bf16: correctness spot checks (host reference), c3- / c4- / c2-like timings, phase breakdown of wave 0
mkdir -p gpurun_out/w32
P=tools/probes/attn_w32_probe.bin
T=tools/probes/attn_w32_probe_timing.bin
{
echo "--- correctness (ragged, masked, tiny, spiked, dk=128)"
timeout 60 $P 5 1 70 192 1 0; timeout 60 $P 7 30 200 192 1 0; timeout 60 $P 9 100 333 192 1 5; timeout 60 $P 3 129 257 192 1 0
timeout 60 $P 6 40 300 128 1 3; timeout 60 $P 4 100 300 192 1 0 3; timeout 60 $P 4 100 300 192 1 0 12; timeout 60 $P 11 33 97 192 1 2
echo "--- c3-like"; timeout 60 $P 64 300 800 192 5 0
echo "--- c2-like"; timeout 60 $P 16 500 900 192 5 0
echo "--- c4-like"; timeout 120 $P 256 250 3300 192 3 0
echo "--- one utterance"; timeout 60 $P 1 650 650 192 5 0
echo "--- phases, c4-like"; timeout 120 $T 256 250 3300 192 2 0
echo "--- phases, c3-like"; timeout 120 $T 64 300 800 192 2 0
} > gpurun_out/w32/time_$1.txt 2>&1
cat gpurun_out/w32/time_$1.txt
