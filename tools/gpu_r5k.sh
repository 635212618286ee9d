#!/bin/bash
# round 5, GPU session K: stream schedules, one process each, three regions per process, two processes per schedule
D=gpurun_out/r5k
mkdir -p $D
: > $D/ab.txt
for wl in c3 c2 c5; do
  K=100; [ $wl = c5 ] && K=50
  for cfg in "0 1" "1 1" "0 2" "1 2" "0 3" "1 3" "0 4"; do
    for rep in 1 2; do
      timeout 120 python tools/debug/stream_schedule_ab.py $wl $cfg $K 2>/dev/null >> $D/ab.txt
    done
  done
done
cat $D/ab.txt
