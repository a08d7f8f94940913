#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box (kernel stats, then FETCH_SIZE and
# WRITE_SIZE in their own runs); results land in gpurun_out/<tag>_{stats,fetch,write}.
#   tools/profile_bench.sh <tag> [bench args...]
set -e
TAG=${1:-prof}
shift || true
ROOT=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_stats -o p -- python3 $ROOT/bench.py --no-cpu-baseline "$@" > $ROOT/gpurun_out/${TAG}_stats.log 2>&1
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/${TAG}_fetch -o p -- python3 $ROOT/bench.py --no-cpu-baseline --steps 60 --warmup 6 > $ROOT/gpurun_out/${TAG}_fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/${TAG}_write -o p -- python3 $ROOT/bench.py --no-cpu-baseline --steps 60 --warmup 6 > $ROOT/gpurun_out/${TAG}_write.log 2>&1
echo "write pass done"
cd $ROOT
tail -1 gpurun_out/${TAG}_stats.log | cut -c1-300
