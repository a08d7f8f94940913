#!/bin/bash
# it/s and ms per launch of bench.py with the given arguments, one line
python bench.py "$@" 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value'],1), round(d['roofline']['avg_launch_ms'],4))"
