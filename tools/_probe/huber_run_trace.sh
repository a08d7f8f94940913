#!/bin/bash
# one run's kernel list of config 4's Huber branch: rocprofv3 --kernel-trace --stats
# over tools/_probe/huber_run_trace.py (4 runs); table in gpurun_out/<tag>_huber_trace
set -e
TAG=${1:-hub}
MIN=${2:-L-BFGS-B}
LOSS=${3:-huber}
ROOT=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_huber_trace -o p -- python3 $ROOT/tools/_probe/huber_run_trace.py 512 4 $MIN $LOSS > $ROOT/gpurun_out/${TAG}_huber_trace.log 2>&1
cat $ROOT/gpurun_out/${TAG}_huber_trace.log
