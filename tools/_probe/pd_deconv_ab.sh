#!/bin/bash
# PD deconvolution at 512^3 with the identity-mode Lanczos halves in the blur on / off
for v in 0 1 0 1; do
  echo -n "lsmr.LANCZOS_IDENTITY=$v: "
  timeout -k 10 300 python tools/bench_pd_deconv.py --set lsmr.LANCZOS_IDENTITY=$v 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d.get('seconds_per_run'))"
done
