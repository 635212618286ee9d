#!/bin/bash
# A/B: s_setprio around the MFMA phases (shipped) against none (library built with -DFS2_SETPRIO=0), interleaved
mkdir -p gpurun_out/r3x
cp fastspeech2_amd/libfs2_hip.so /tmp/prio.so
for rep in 0 1; do for v in prio noprio; do
  if [ $v = noprio ]; then cp tools/tmp_ab/libfs2_hip_noprio.so fastspeech2_amd/libfs2_hip.so; else cp /tmp/prio.so fastspeech2_amd/libfs2_hip.so; fi
  for wl in c3 c4; do python bench.py --no-cpu-baseline --workload $wl > gpurun_out/r3x/bench_${wl}_${v}_$rep.json 2>/dev/null; done
done; done
cp /tmp/prio.so fastspeech2_amd/libfs2_hip.so
for f in gpurun_out/r3x/bench_*.json; do echo "$f $(python -c "import json,sys;d=json.load(open('$f'));print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'])")"; done
