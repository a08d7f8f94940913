#!/usr/bin/env python3
"""Cost of reading one device scalar back after a small kernel, three ways:
tensor.cpu(), a non-blocking copy into pinned memory + event wait, and a result
slot that IS pinned host memory (the kernel stores through its device address)
+ stream synchronize."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nsol_amd import _lib, ops  # noqa: E402
from nsol_amd.device import stream_ptr  # noqa: E402

n = 1 << 16
x = torch.rand(n, device="cuda")
ws, res = ops._workspace(x.device)
lib = _lib.load()
pinned = torch.zeros(8, dtype=torch.float64).pin_memory()
host = pinned.numpy()
ev = torch.cuda.Event()
big = torch.rand(1 << 27, device="cuda")


def launch(result_ptr):
    lib.nsol_dot_f32(x.data_ptr(), x.data_ptr(), n, result_ptr, ws.data_ptr(), stream_ptr())


def way_cpu():
    launch(res.data_ptr())
    return float(res.cpu()[0])


def way_item():
    launch(res.data_ptr())
    return float(res.item())


def way_pinned_copy():
    launch(res.data_ptr())
    pinned[:1].copy_(res, non_blocking=True)
    ev.record()
    ev.synchronize()
    return float(host[0])


def way_pinned_slot():
    launch(pinned.data_ptr())
    torch.cuda.current_stream().synchronize()
    return float(host[0])


def way_pinned_slot_event():
    launch(pinned.data_ptr())
    ev.record()
    ev.synchronize()
    return float(host[0])


ref = way_cpu()
for name, fn in (("cpu()", way_cpu), ("item()", way_item), ("pinned copy + event", way_pinned_copy),
                 ("pinned slot + stream sync", way_pinned_slot),
                 ("pinned slot + event", way_pinned_slot_event)):
    assert fn() == ref, (name, fn(), ref)
    for with_big in (False, True):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 300
        for _ in range(reps):
            if with_big:
                ops.dot(big[:1 << 22], big[:1 << 22]) if False else None
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps * 1e6
        if not with_big:
            print("%-28s %.1f us per launch + read-back" % (name, dt), flush=True)
