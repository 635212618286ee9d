#!/usr/bin/env python3
"""Summarise the rocprofv3 output of tools/profile_round.sh (gpurun_out/prof_<tag>/) into profiles/:
  <tag>_kernel_stats.csv   per-kernel totals / averages of the --kernel-trace --stats run
  <tag>_pmc_summary.csv    per (kernel, grid) averages of every PMC counter collected
  <tag>_traffic.json       HBM traffic per launch of the dominant kernel (bench.py's roofline.traffic)
HBM traffic = 2 * FETCH_SIZE + WRITE_SIZE (KB -> bytes): on gfx950 FETCH_SIZE tallies the 128-byte requests of wide
coalesced reads at 64 B (MI355X_MICROARCH.md, HBM/rocprofv3 section); WRITE_SIZE is used as is."""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
workload = os.environ.get("FS2_PROF_WORKLOAD", "c3")
precision = os.environ.get("FS2_PROF_PRECISION", "mix_mx4")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")

def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]

stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, tag + "_kernel_stats.csv"))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("fetch", "write", "sq", "sq2"):
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if sub == "sq" and r["Counter_Name"] == "SQ_WAVES":
                agg[k]["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
if not agg:
    print("no PMC passes under", src, "(kernel stats copied)")
    sys.exit(0)
names = sorted({c for v in agg.values() for c in v})
rows = []
for (kern, grid), v in agg.items():
    n = max(len(x) for x in v.values())
    rows.append([kern.replace(",", ";"), grid, n] + [("%.6g" % (sum(v[c]) / len(v[c])) if v.get(c) else "") for c in names])
rows.sort(key=lambda r: -(float(r[3 + names.index("dur_us")] or 0) * r[2]) if "dur_us" in names else 0)
with open(os.path.join(dst, tag + "_pmc_summary.csv"), "w") as f:
    f.write("kernel,grid_size,dispatches," + ",".join(names) + "\n")
    for r in rows:
        f.write(",".join(str(x) for x in r) + "\n")
# dominant kernel = largest total duration in the SQ pass
dom = max(agg.items(), key=lambda kv: sum(kv[1].get("dur_us", [0])))
(kern, grid), v = dom
fetch = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"])
write = sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
info = {"_comment": "HBM-side traffic of the dominant kernel per launch: rocprofv3 PMC passes (tools/profile_round.sh) of `python bench.py --workload %s --precision %s "
                    "--steps 4 --warmup 2 --no-cpu-baseline` on MI355X;" % (workload, precision) + " FETCH_SIZE and WRITE_SIZE from separate --pmc runs, averaged over all launches of the kernel/grid; "
                    "traffic = 2*FETCH_SIZE + WRITE_SIZE (gfx950 correction, MI355X_MICROARCH.md). Raw table: profiles/%s_pmc_summary.csv" % tag,
        "workload": workload, "precision": precision, "kernel": kern, "grid_size": grid, "launches_averaged": len(v["FETCH_SIZE"]),
        "fetch_size_kb_raw": round(fetch, 1), "write_size_kb": round(write, 1), "traffic_bytes": int((2 * fetch + write) * 1024),
        "avg_duration_us_pmc_run": round(sum(v["dur_us"]) / len(v["dur_us"]), 1)}
sys.path.insert(0, root)
import bench      # noqa: E402  (csrc_sha16: the fingerprint of the kernel sources this record was measured on)
info["csrc_sha16"] = bench.csrc_sha16()
if len(sys.argv) > 2:
    info["kernel_site"] = sys.argv[2]
# matrix-pipe occupancy and wave-state shares of the same kernel (the algorithmic byte count is bench.py's business: it computes it from
# the launch's shapes and reports traffic / algorithmic itself)
def avg(c):
    return sum(v[c]) / len(v[c]) if v.get(c) else None
if avg("SQ_VALU_MFMA_BUSY_CYCLES") and avg("SQ_BUSY_CU_CYCLES"):
    info["mfma_busy"] = round(avg("SQ_VALU_MFMA_BUSY_CYCLES") / (4.0 * avg("SQ_BUSY_CU_CYCLES")), 3)
if avg("SQ_WAVE_CYCLES"):
    for c, k in (("SQ_WAIT_ANY", "waves_parked"), ("SQ_WAIT_INST_ANY", "waves_issue_stalled"), ("SQ_ACTIVE_INST_ANY", "waves_issuing")):
        if avg(c) is not None:
            info[k] = round(avg(c) / avg("SQ_WAVE_CYCLES"), 3)
if avg("SQ_LDS_IDX_ACTIVE"):
    info["lds_bank_conflict_share"] = round((avg("SQ_LDS_BANK_CONFLICT") or 0.0) / avg("SQ_LDS_IDX_ACTIVE"), 4)
if avg("SQ_INSTS_MFMA"):
    info["valu_per_mfma"] = round(((avg("SQ_INSTS_VALU") or 0.0) - avg("SQ_INSTS_MFMA")) / avg("SQ_INSTS_MFMA"), 2)
json.dump(info, open(os.path.join(dst, tag + "_traffic.json"), "w"), indent=1)
print(json.dumps(info, indent=1))
