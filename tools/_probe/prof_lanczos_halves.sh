#!/bin/bash
# rocprofv3 passes over tools/_probe/run_lanczos_halves.py: stats, traffic, SQ activity.
#   tools/_probe/prof_lanczos_halves.sh <tag>
set -e
TAG=${1:-lz}
ROOT=$(pwd)
O=$ROOT/gpurun_out/${TAG}_lzprof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  rocprofv3 "$@" --kernel-trace --output-format csv -d $O/$name -o p -- python3 $ROOT/tools/_probe/run_lanczos_halves.py 512 6 > $O/$name.log 2>&1 || echo "pass $name failed"; }
run stats --stats
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run sq1 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS
run sq2 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA
run tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
cd $ROOT
python3 tools/summarize_pmc.py k_blur3_dma $O/stats $O/fetch $O/write $O/sq1 $O/sq2 $O/tcc > gpurun_out/${TAG}_lzprof.jsonl
