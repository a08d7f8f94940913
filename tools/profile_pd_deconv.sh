#!/bin/bash
# rocprofv3 --kernel-trace --stats over tools/bench_pd_deconv.py;
# results in gpurun_out/<tag>_pd_deconv.
set -e
TAG=${1:-prof}
ROOT=$(pwd)
mkdir -p gpurun_out
python3 tools/bench_pd_deconv.py > gpurun_out/${TAG}_bench_pd_deconv.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_pd_deconv -o p -- python3 $ROOT/tools/bench_pd_deconv.py --repeat 2 > $ROOT/gpurun_out/${TAG}_pd_deconv.log 2>&1
echo "pd deconv done"
