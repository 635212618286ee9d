#!/usr/bin/env python3
"""rocprofv3 --kernel-trace of bench.py (tools/profile_round.sh): dispatch durations of the dominant kernel by region of the run -- the timed region
(steps issued on several streams) and the one-stream region behind it -- and the average number of kernels in flight in each.
  python tools/trace_regions.py gpurun_out/prof_r05/trace 'gemm_pl_bf16<1, 256, false, 2>' [timed steps = 4]"""
import csv, glob, os, sys

src, kern = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
f = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
dom = [r for r in rows if kern in r["Kernel_Name"]]
side = [i for i, r in enumerate(dom) if r["Stream_Id"] != "0"]
# the timed steps are the LAST `steps` steps issued on the step streams; launches per step = launches on the null stream's first (synchronous) call
first_null = 0
while first_null < len(dom) and dom[first_null]["Stream_Id"] == "0":
    first_null += 1
lps = first_null
timed = [dom[i] for i in side[-steps * lps:]]
alone = [r for r in dom[side[-1] + 1:]][2 * lps:]          # (behind two untimed steps)


def region(name, ds):
    if not ds:
        return
    t0, t1 = int(ds[0]["Start_Timestamp"]), int(ds[-1]["End_Timestamp"])
    reg = [r for r in rows if int(r["Start_Timestamp"]) >= t0 and int(r["End_Timestamp"]) <= t1]
    busy = sum(dur(r) for r in reg)
    d = [dur(r) / 1e3 for r in ds]
    print("%-28s %3d dispatches of the kernel on streams %s: avg %.1f us (min %.1f, max %.1f); all kernels between its first start and last end: "
          "%d dispatches, %.3f ms of kernel time in %.3f ms = %.2f kernels in flight"
          % (name, len(ds), sorted({r["Stream_Id"] for r in ds}), sum(d) / len(d), min(d), max(d), len(reg), busy / 1e6, (t1 - t0) / 1e6, busy / (t1 - t0)))


print("kernel:", kern, "| launches per step:", lps)
region("timed region (step streams)", timed)
region("one-stream region", alone)
