#!/bin/bash
# timing probe: cross-term units of the mx conv as block-scaled fp6 (e2m3) instead of e4m3
mkdir -p gpurun_out/r3p
P=tools/probes/mx_conv_probe.bin
{
for rep in 1 2; do
echo "--- mx (e4m3 cross terms), c3-sized"; $P 36352 384 1536 9 256 2 | tail -2
echo "--- fp6 probe, c3-sized"; timeout 120 $P 36352 384 1536 9 2256 2 | tail -2
done
echo "--- split-bf16 x3, c3-sized"; $P 36352 384 1536 9 256 0 | tail -1
echo "--- mx, c4-sized"; $P 460000 384 1536 9 256 2 | tail -2
echo "--- fp6 probe, c4-sized"; timeout 120 $P 460000 384 1536 9 2256 2 | tail -2
} > gpurun_out/r3p/fp6_probe.txt 2>&1
cat gpurun_out/r3p/fp6_probe.txt | cut -c1-200
