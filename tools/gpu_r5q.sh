#!/bin/bash
# round 5, GPU session Q: kernel-choice options under three steps in flight (they were tuned for one step at a time)
D=gpurun_out/r5q
mkdir -p $D
: > $D/ab.txt
for wl in c3 c2; do
  for opt in "" "FS2_NOSPLITK=1" "FS2_ROW8=1" "FS2_ROW8=1 FS2_QKV8=1" "FS2_BAL=1" "FS2_BAL=2" "FS2_BM=128" "FS2_BM=256" "FS2_QKV_SPLIT=0" "FS2_QKV_SPLIT=1" "FS2_FUSE_VAR=0" "FS2_MT4=4" "FS2_NOSPLITK=1 FS2_ROW8=1 FS2_QKV8=1"; do
    echo -n "[$opt] " >> $D/ab.txt
    env $opt timeout 120 python tools/debug/stream_schedule_ab.py $wl 0 3 100 2>/dev/null >> $D/ab.txt
  done
done
cat $D/ab.txt
