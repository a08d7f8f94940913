#!/bin/bash
python bench.py --no-cpu-baseline --no-config4 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['roofline']['avg_launch_ms'],4))"
