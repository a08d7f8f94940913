#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (separate runs, kernel trace only) over
# bench_admm.py, L-BFGS-B / Huber branch; results in
# gpurun_out/<tag>_huber_pmc_{fetch,write}.
set -e
TAG=${1:-prof}
ROOT=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/${TAG}_huber_pmc_fetch -o p -- python3 $ROOT/bench_admm.py --minimizer L-BFGS-B --data-loss huber --no-cpu-baseline --repeat 1 > $ROOT/gpurun_out/${TAG}_huber_pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/${TAG}_huber_pmc_write -o p -- python3 $ROOT/bench_admm.py --minimizer L-BFGS-B --data-loss huber --no-cpu-baseline --repeat 1 > $ROOT/gpurun_out/${TAG}_huber_pmc_write.log 2>&1
echo "write done"
