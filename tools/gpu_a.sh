#!/bin/bash
mkdir -p gpurun_out/r3p
P=tools/probes/mx_conv_probe.bin; T=tools/probes/mx_conv_probe_t.bin
{
echo "--- baseline 4-wave BM 256"; $P 36352 384 1536 9 256 2 | tail -2; $P 36352 384 1536 9 256 0 | tail -2
for bm in 1192 1160 1128; do
echo "--- ping-pong $bm, fragments fetched in the DMA phase"; timeout 120 $P 36352 384 1536 9 $bm 2 | tail -3; timeout 120 $P 36352 384 1536 9 $bm 0 | tail -3
done
echo "--- phases (timing build)"; timeout 120 $T 36352 384 1536 9 1160 2 | tail -5; timeout 120 $T 36352 384 1536 9 1192 0 | tail -5; timeout 120 $T 36352 384 1536 9 1128 2 | tail -5
echo "--- c4-sized"; $P 460000 384 1536 9 256 2 | tail -2; timeout 120 $P 460000 384 1536 9 1160 2 | tail -3; $P 460000 384 1536 9 256 0 | tail -2; timeout 120 $P 460000 384 1536 9 1192 0 | tail -3
} > gpurun_out/r3p/pp_probe4.txt 2>&1
cat gpurun_out/r3p/pp_probe4.txt | cut -c1-250
