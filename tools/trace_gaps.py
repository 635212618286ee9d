#!/usr/bin/env python3
"""GPU busy / idle time per bench step from a rocprofv3 --kernel-trace CSV (tools/profile_round.sh):
steps are delimited by the fs2::embed_pe launches; prints per-step span, kernel-time sum, idle time and the largest gaps."""
import csv, glob, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
f = glob.glob(os.path.join(root, "gpurun_out", "prof_" + tag, "trace", "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].replace("void ", "").split("(")[0]
marks = [i for i, r in enumerate(rows) if "embed_pe" in name(r)]
for a, b in zip(marks[:-1], marks[1:]):
    seg = rows[a:b]
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    gaps = []
    for x, y in zip(seg[:-1], seg[1:] + []):
        g = int(y["Start_Timestamp"]) - int(x["End_Timestamp"])
        if g > 2000:
            gaps.append((g / 1e3, name(x), name(y)))
    tail = (t1 - int(seg[-1]["End_Timestamp"])) / 1e3
    print("step: span %.1f us, kernels %.1f us (%d launches), idle %.1f us (tail to next step %.1f us)" % ((t1 - t0) / 1e3, busy / 1e3, len(seg), (t1 - t0 - busy) / 1e3, tail))
    for g, x, y in sorted(gaps, reverse=True)[:4]:
        print("      gap %.1f us between %s and %s" % (g, x[:40], y[:40]))
