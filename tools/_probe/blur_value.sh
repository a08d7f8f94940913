#!/bin/bash
python tools/bench_blur3.py 512 2>/dev/null | grep -v amdgpu | head -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms'], d['min_ms'])"
