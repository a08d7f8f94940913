#!/bin/bash
# rocprofv3 passes over the one-pass blur (tools/run_blur3.py): kernel stats,
# FETCH_SIZE, WRITE_SIZE, and SQ activity counters, each in its own run.
# Results under gpurun_out/<tag>_blur3_*.
set -e
TAG=${1:-prof}
ROOT=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
run() {  # name, rocprof args...
  local name=$1; shift
  rocprofv3 "$@" --kernel-trace --output-format csv -d $ROOT/gpurun_out/${TAG}_blur3_${name} -o p -- python3 $ROOT/tools/run_blur3.py 512 20 1 > $ROOT/gpurun_out/${TAG}_blur3_${name}.log 2>&1
  echo "$name done"
}
run stats --stats
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run sq1 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS
run sq2 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_WAVES
