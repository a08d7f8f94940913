"""Adapters between the solvers' device-resident state and caller-supplied
callables (the reference's operator contract: flat array in, flat array out).

nsol_amd operators / proxes accept torch HIP tensors, so caller lambdas built
from them (reshape -> operator -> flatten) run entirely on the GPU.  A foreign
callable that only understands NumPy gets its argument copied to the host and
its result copied back; the solver state itself never leaves HBM.
"""
import numpy as np

from ._lib import NsolHipError
from .device import is_device_tensor, to_device, to_numpy


def _is_gpu_failure(exc):
    """A failing HIP launch / allocation, as opposed to a NumPy-only callable
    choking on a tensor argument."""
    if isinstance(exc, NsolHipError):
        return True
    import torch
    if isinstance(exc, getattr(torch, "OutOfMemoryError", ())):
        return True
    if isinstance(exc, getattr(torch, "AcceleratorError", ())):
        return True
    msg = str(exc)
    return isinstance(exc, RuntimeError) and (
        "HIP" in msg or "hip" in msg or "CUDA" in msg or
        "out of memory" in msg)


class BridgedCallable(object):

    def __init__(self, fn, dtype):
        self.fn = fn
        self.dtype = dtype
        self.on_device = None      # unknown until the first call

    def _host_call(self, t, *args):
        res = self.fn(to_numpy(t), *args)
        return to_device(np.asarray(res, dtype=np.float64).reshape(-1),
                         self.dtype)

    @staticmethod
    def _checked(out, like):
        """A device callable hands back raw memory the kernels index by the
        first operand's type: a result of another dtype is a caller error."""
        if out.dtype != like.dtype:
            raise ValueError(
                "operator returned %s for a %s solve: build the operators / "
                "pass dtype= consistently" % (out.dtype, like.dtype))
        return out.contiguous().view(-1)

    def __call__(self, t, *args):
        if self.on_device is None:
            out = None
            try:
                out = self.fn(t, *args)
            except (TypeError, AttributeError, ValueError, RuntimeError) as e:
                # a NumPy-only callable choking on a device tensor takes the
                # host bridge (a genuine error of its own re-surfaces from the
                # host call below); a GPU failure must not be hidden behind it
                if _is_gpu_failure(e):
                    raise
            if is_device_tensor(out):
                self.on_device = True
                return self._checked(out, t)
            self.on_device = False
            return self._host_call(t, *args)
        if self.on_device:
            return self._checked(self.fn(t, *args), t)
        return self._host_call(t, *args)
