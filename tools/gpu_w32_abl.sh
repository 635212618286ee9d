#!/bin/bash
# ablations of attn_w32 (timing only): which part of the tile costs what.  c4-like and c3-like.
mkdir -p gpurun_out/w32
{
echo "--- full kernel"; timeout 120 tools/probes/attn_w32_probe.bin 256 250 3300 192 2 0 | cut -c1-160
for a in 1 2 4 8 3 11 15; do echo "--- ablation mask $a (1: no DMA, 2: no softmax in the blocks, 4: no closing wait/barrier, 8: no fragment reads)"; timeout 120 tools/probes/attn_w32_probe_abl$a.bin 256 250 3300 192 2 0 | cut -c1-160; done
} > gpurun_out/w32/ablations.txt 2>&1
cat gpurun_out/w32/ablations.txt
