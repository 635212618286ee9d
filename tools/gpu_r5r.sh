#!/bin/bash
# round 5, GPU session R: kernel trace + stats of the one-stream schedule (every kernel alone on the chip), and of the default command, on ONE box
D=gpurun_out/r5r
mkdir -p $D
export TMPDIR=/tmp
R=/root/repo
cd /tmp
CMD="python $R/bench.py --workload c3 --precision mix_mx --steps 20 --warmup 5 --no-cpu-baseline --sustain 0 --streams 1 --no-overlap-encoder"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$D/trace1 -o t -- $CMD > $R/$D/trace1.log 2>&1
CMD="python $R/bench.py --workload c3 --precision mix_mx --steps 20 --warmup 5 --no-cpu-baseline --sustain 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$D/trace3 -o t -- $CMD > $R/$D/trace3.log 2>&1
head -3 $R/$D/trace1/t_kernel_stats.csv | cut -c1-150; head -3 $R/$D/trace3/t_kernel_stats.csv | cut -c1-150
tail -n 1 $R/$D/trace1.log | cut -c1-300
