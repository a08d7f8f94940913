#!/bin/bash
python bench_admm.py --size 511 --no-cpu-baseline --repeat 3 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['seconds_per_run'],4))"
