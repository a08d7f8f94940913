"""In-run durations of the C-ABI entries (measurement only: bench.py, bench_admm.py).

    with _timing.KernelTimer() as kt:
        solver.run()
    kt.summary()   ->  {"corr3_wrap_lanczos_b2": {"launches": 100, "avg_ms": .., ..}, ..}

While a timer is active every entry fetched through ops._fn / lbfgsb_device._fn is
bracketed by two events on the stream it is launched on (torch's current stream, the
one `stream_ptr()` hands to the library), so the figures are those of the kernels as
they run INSIDE the solve -- cold caches, the neighbours they really have, the
clocks of the moment -- not of a replay on idle operands.  An entry that launches a
large kernel and a one-workgroup reduction behind it (the Lanczos halves, the
reductions) is timed as a whole.  The events cost a few microseconds per entry: a run
under a timer is slower than the run that is reported as seconds_per_run.
"""
import torch

_active = None


def active():
    return _active


class KernelTimer(object):

    def __init__(self, tags=None):
        """tags: {entry name: index of an integer argument} -- the entry is then kept
        per value of that argument as "name#value" (the number of stored vectors a
        limited-memory product runs over decides its bytes)."""
        self.records = []               # (name, start event, end event)
        self.tags = dict(tags or {})

    def __enter__(self):
        global _active
        self._outer = _active
        _active = self
        return self

    def __exit__(self, *exc):
        global _active
        _active = self._outer
        return False

    def wrap(self, name, fn):
        records = self.records
        tag = self.tags.get(name)

        def timed(*args):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            records.append((name if tag is None else "%s#%d" % (name, int(args[tag])),
                            e0, e1))
            return rc
        return timed

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1 in self.records:
            ms = e0.elapsed_time(e1)
            s = out.setdefault(name, {"launches": 0, "total_ms": 0.0,
                                      "min_ms": ms, "max_ms": ms})
            s["launches"] += 1
            s["total_ms"] += ms
            s["min_ms"] = min(s["min_ms"], ms)
            s["max_ms"] = max(s["max_ms"], ms)
        for s in out.values():
            s["avg_ms"] = s["total_ms"] / s["launches"]
        return out
