"""Adapters between the solvers' device-resident state and caller-supplied
callables (the reference's operator contract: flat array in, flat array out).

nsol_amd operators / proxes accept torch HIP tensors, so caller lambdas built
from them (reshape -> operator -> flatten) run entirely on the GPU.  A foreign
callable that only understands NumPy gets its argument copied to the host and
its result copied back; the solver state itself never leaves HBM.
"""
import numpy as np

from .device import is_device_tensor, to_device, to_numpy


class BridgedCallable(object):

    def __init__(self, fn, dtype):
        self.fn = fn
        self.dtype = dtype
        self.on_device = None      # unknown until the first call

    def _host_call(self, t, *args):
        res = self.fn(to_numpy(t), *args)
        return to_device(np.asarray(res, dtype=np.float64).reshape(-1),
                         self.dtype)

    def __call__(self, t, *args):
        if self.on_device is None:
            try:
                out = self.fn(t, *args)
                if is_device_tensor(out):
                    self.on_device = True
                    return out.contiguous().view(-1)
            except (TypeError, AttributeError, ValueError, RuntimeError):
                # a NumPy-only callable choking on a device tensor; a genuine
                # error re-surfaces from the host call below
                pass
            self.on_device = False
            return self._host_call(t, *args)
        if self.on_device:
            return self.fn(t, *args).contiguous().view(-1)
        return self._host_call(t, *args)
