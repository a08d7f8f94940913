#!/bin/bash
# rocprofv3 --kernel-trace --stats over bench_admm.py (LSMR branch, then the
# L-BFGS-B / Huber branch); results in gpurun_out/<tag>_admm_{lsmr,lbfgsb}.
set -e
TAG=${1:-prof}
ROOT=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_admm_lsmr -o p -- python3 $ROOT/bench_admm.py > $ROOT/gpurun_out/${TAG}_admm_lsmr.log 2>&1
echo "lsmr done"
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_admm_lbfgsb -o p -- python3 $ROOT/bench_admm.py --minimizer L-BFGS-B --data-loss huber > $ROOT/gpurun_out/${TAG}_admm_lbfgsb.log 2>&1
echo "lbfgsb done"
