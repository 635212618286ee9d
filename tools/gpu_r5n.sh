#!/bin/bash
# round 5, GPU session N: kernel trace of the ONE-stream schedule (every kernel alone on the chip) beside the default command's, and the RCCL world-size-1 line
D=gpurun_out/r5n
mkdir -p $D
export TMPDIR=/tmp
R=/root/repo
cd /tmp
CMD="python $R/bench.py --workload c3 --precision mix_mx --steps 4 --warmup 2 --no-cpu-baseline --sustain 0 --streams 1 --no-overlap-encoder"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$D/trace1 -o t -- $CMD > $R/$D/trace1.log 2>&1
CMD="python $R/bench.py --workload c3 --precision mix_mx --steps 20 --warmup 5 --no-cpu-baseline --sustain 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$D/trace20 -o t -- $CMD > $R/$D/trace20.log 2>&1
cd $R
FS2_FORCE_DIST=1 python bench.py --workload c5 --no-cpu-baseline > $D/bench_r05_c5_rccl_single_rank.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > $D/bench_driver_form.json 2>$D/bench_driver_form.err
tail -c 600 $D/bench_driver_form.json; ls $D/trace1 $D/trace20
