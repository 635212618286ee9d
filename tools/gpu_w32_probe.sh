#!/bin/bash
# attn_w32 against attn_bf16 on the GPU: correctness on ragged / masked / tiny cases (host reference), then c3- and c4-like timings
mkdir -p gpurun_out/w32
P=tools/probes/attn_w32_probe.bin
{
echo "--- tiny / ragged"; timeout 60 $P 5 1 70 192 1 0
timeout 60 $P 7 30 200 192 1 0
timeout 60 $P 9 100 333 192 1 5
timeout 60 $P 3 129 257 192 1 0
timeout 60 $P 6 40 300 128 1 3
echo "--- spiked keys (deferred maximum / slow rows)"; timeout 60 $P 4 100 300 192 1 0 3
timeout 60 $P 4 100 300 192 1 0 12
timeout 60 $P 4 100 300 192 1 0 60
echo "--- c3-like (64 utterances, ~550 frames)"; timeout 60 $P 64 300 800 192 5 0
echo "--- c2-like"; timeout 60 $P 16 500 900 192 5 0
echo "--- c4-like (256 utterances, 250-3300 frames)"; timeout 120 $P 256 250 3300 192 3 0
echo "--- one utterance"; timeout 60 $P 1 650 650 192 5 0
echo "--- encoder-like dk=128"; timeout 60 $P 64 60 130 128 5 0
} > gpurun_out/w32/probe.txt 2>&1
cat gpurun_out/w32/probe.txt
