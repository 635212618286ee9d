#!/bin/bash
# round 5, GPU session M: more hardware queues for the stream schedules?
D=gpurun_out/r5m
mkdir -p $D
: > $D/ab.txt
for q in 4 8; do
  echo "GPU_MAX_HW_QUEUES=$q" >> $D/ab.txt
  for wl in c3 c2; do
    for cfg in "0 3" "0 4" "0 6" "1 3"; do
      GPU_MAX_HW_QUEUES=$q timeout 120 python tools/debug/stream_schedule_ab.py $wl $cfg 100 2>/dev/null >> $D/ab.txt
    done
  done
done
cat $D/ab.txt
