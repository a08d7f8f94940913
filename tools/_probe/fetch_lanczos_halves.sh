#!/bin/bash
# FETCH_SIZE of the blur's forms with the installed library: tools/_probe/fetch_lanczos_halves.sh <tag>
set -e
TAG=${1:-lzf}
ROOT=$(pwd)
O=$ROOT/gpurun_out/${TAG}_lzfetch
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o p -- python3 $ROOT/tools/_probe/run_lanczos_halves.py 512 6 > $O/fetch.log 2>&1 || echo "pass failed"
cd $ROOT
python3 tools/summarize_pmc.py k_blur3_dma $O/fetch | grep FETCH | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['kernel'][:60], 'FETCH_SIZE mean', d['mean'], '-> %.2f B/voxel (x2 x 1 KiB units / 2^27)' % (2 * d['mean'] * 1024 / 2**27))
"
