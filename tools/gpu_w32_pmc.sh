#!/bin/bash
# PMC passes over the attention probe (c4-like): attn_bf16 and attn_w32 side by side.  usage: gpu_w32_pmc.sh <tag> [W32_DMA=1 in env for the DMA variant]
TAG=${1:-w32}
OUT=/root/repo/gpurun_out/w32/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
P=/root/repo/tools/probes/attn_w32_probe.bin
cd /tmp
ARGS="256 250 3300 192 2 0"
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq -o p -- $P $ARGS > $OUT/sq.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $OUT/sq2 -o p -- $P $ARGS > $OUT/sq2.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/sq3 -o p -- $P $ARGS > $OUT/sq3.log 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "attn" not in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for v in agg.values() for c in v})
print("kernel," + ",".join(names))
for k, v in agg.items():
    print(k.replace(",", ";") + "," + ",".join("%.6g" % (sum(v[c]) / len(v[c])) if v.get(c) else "" for c in names))
PY
