#!/bin/bash
# PMC passes over tools/tune_pdk.py for one configuration (run on the GPU box):
#   tools/prof_pdk.sh <cfg> <outdir>
set -e
CFG=${1:-3:12:8:0}
OUT=${2:-gpurun_out/pdk_prof}
ROOT=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $ROOT/$OUT/$tag -o p -- python3 $ROOT/tools/tune_pdk.py --cfg $CFG --rounds 1 --launches 3 > $ROOT/$OUT/$tag.log 2>&1 || echo "pass $tag failed"
done
cd $ROOT
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_pd_fused" in k:
            acc[(k[:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print("%-62s %-22s n=%d mean=%.4g" % (k, c, len(v), sum(v) / len(v)))
PY
