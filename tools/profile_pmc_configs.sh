#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes of bench.py pinned to each given kernel
# configuration (waves:ntx:zchunk ...); results in gpurun_out/pmc_<cfg>_{fetch,write}.
set -e
ROOT=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for CFG in "$@"; do
  T=$(echo $CFG | tr ':' '_')
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_${T}_fetch -o p -- python3 $ROOT/bench.py --no-cpu-baseline --no-verify --no-config4 --pdk $CFG --steps 60 --warmup 6 > $ROOT/gpurun_out/pmc_${T}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_${T}_write -o p -- python3 $ROOT/bench.py --no-cpu-baseline --no-verify --no-config4 --pdk $CFG --steps 60 --warmup 6 > $ROOT/gpurun_out/pmc_${T}_write.log 2>&1
  echo "done $CFG"
done
