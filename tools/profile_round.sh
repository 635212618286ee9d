#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 passes over bench steps.  Kernel trace + stats in one run, each PMC group
# in its own run (no trace domains together with --pmc).  Raw output -> gpurun_out/prof_<tag>/, summarise with
# tools/pmc_summary.py on the build host and commit the summaries under profiles/.
#   tools/profile_round.sh <tag> [workload] [precision] [pmc: 0|1] ["bench.py arguments" instead of the default --steps 4 --warmup 2]
set -u
TAG=${1:-r06}
WL=${2:-c3}
PREC=${3:-mix_mx4}
PMC=${4:-1}
ARGS=${5:---steps 4 --warmup 2}
OUT=/root/repo/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
CMD="python /root/repo/bench.py --workload $WL --precision $PREC $ARGS --no-cpu-baseline --sustain 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- $CMD > "$OUT/trace.log" 2>&1
if [ "$PMC" = "1" ]; then
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o p -- $CMD > "$OUT/fetch.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o p -- $CMD > "$OUT/write.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/sq" -o p -- $CMD > "$OUT/sq.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d "$OUT/sq2" -o p -- $CMD > "$OUT/sq2.log" 2>&1
fi
find "$OUT" -name "*.csv" | head -20
