#!/usr/bin/env python3
"""Config 4's Huber branch against lbfgsb_device.WINDOW_STRETCH (run time, number of
windows fetched by the Cauchy searches)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from huber_run_trace import factory  # noqa: E402
from nsol_amd import lbfgsb, lbfgsb_device  # noqa: E402

make = factory(int(sys.argv[1]) if len(sys.argv) > 1 else 512)
make().run()
ref = None
for stretch in (1.0, 1.25, 1.5, 2.0, 3.0, 1.0):
    lbfgsb_device.WINDOW_STRETCH = stretch
    ts = []
    for _ in range(4):
        for k in lbfgsb.STATS:
            lbfgsb.STATS[k] = 0
        s = make()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.run()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    x = s.get_x_device()
    if ref is None:
        ref = x.clone()
    print("stretch %.2f: %.4f s (min %.4f), fetches %d, same result: %s"
          % (stretch, float(np.median(ts)), min(ts), lbfgsb.STATS["fetches"],
             bool(torch.equal(x, ref))), flush=True)
