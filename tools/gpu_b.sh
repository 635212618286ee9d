#!/bin/bash
mkdir -p gpurun_out/r3c
cd tools/probes
{
echo "--- before (st0)"; for ar in 2 0; do ./mx_conv_probe_st0.bin 30208 384 1024 9 256 $ar 1 | tail -1 | cut -c1-90; done
echo "--- A-address hoist"; for ar in 2 0; do for bm in 256 128 64; do ./mx_conv_probe_addr.bin 30208 384 1024 9 $bm $ar 1 | tail -1 | cut -c1-90; done; done
echo "--- c4 size"; for ar in 2 0; do ./mx_conv_probe_addr.bin 376832 384 1024 9 256 $ar 0 | tail -1 | cut -c1-90; done
} > ../../gpurun_out/r3c/addr_probe.txt 2>&1
cat ../../gpurun_out/r3c/addr_probe.txt
