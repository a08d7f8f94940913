#!/bin/bash
# config 4 (LSMR branch), both halves of a Lanczos step in the blur against blur + blur +
# update, three interleaved rounds on one box: seconds per 10 x 10 run
for r in 1 2 3; do
  for v in 1 0; do
    python bench_admm.py --no-cpu-baseline --repeat 5 --set lsmr.USE_BLUR_LANCZOS=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in_blur=$v', round(d['seconds_per_run'],4), d['rel_change_vs_input'])"
  done
done
