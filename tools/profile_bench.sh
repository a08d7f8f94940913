#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box: kernel stats of the default
# (self-tuning) run, then FETCH_SIZE and WRITE_SIZE in their own runs, pinned to
# the configuration the first run settled on.  Results land in
# gpurun_out/<tag>_{stats,fetch,write}; gpurun_out/<tag>_config holds the config.
#   tools/profile_bench.sh <tag> [bench args...]
set -e
TAG=${1:-prof}
shift || true
ROOT=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_stats -o p -- python3 $ROOT/bench.py --no-cpu-baseline "$@" > $ROOT/gpurun_out/${TAG}_stats.log 2>&1
echo "stats pass done"
CFG=$(python3 - <<PY
import json
for line in open("$ROOT/gpurun_out/${TAG}_stats.log"):
    if line.startswith('{"metric"'):
        c = json.loads(line)["config"]["kernel_config"]
        print("%d:%d:%d" % (c["waves"], c["tiles_x"], c["zchunk"]))
PY
)
echo $CFG > $ROOT/gpurun_out/${TAG}_config
echo "config $CFG"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/${TAG}_fetch -o p -- python3 $ROOT/bench.py --no-cpu-baseline --no-verify --no-config4 --pdk $CFG --steps 60 --warmup 6 > $ROOT/gpurun_out/${TAG}_fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/${TAG}_write -o p -- python3 $ROOT/bench.py --no-cpu-baseline --no-verify --no-config4 --pdk $CFG --steps 60 --warmup 6 > $ROOT/gpurun_out/${TAG}_write.log 2>&1
echo "write pass done"
