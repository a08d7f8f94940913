"""Device plumbing: PyTorch-ROCm is used only for HBM allocations, H2D/D2H
copies, streams and torch.distributed -- all arithmetic goes through
libnsol_hip.so (see ops.py)."""
import numpy as np
import torch

_default_dtype = np.float32


def set_default_dtype(dtype):
    """Working precision of the solvers (np.float32 default; np.float64 for
    validation against the float64 reference)."""
    global _default_dtype
    dtype = np.dtype(dtype).type
    if dtype not in (np.float32, np.float64):
        raise ValueError("dtype must be float32 or float64")
    _default_dtype = dtype


def get_default_dtype():
    return _default_dtype


def torch_dtype(dtype):
    return torch.float32 if np.dtype(dtype) == np.float32 else torch.float64


def suffix(t):
    """C-ABI suffix for a device tensor."""
    if t.dtype == torch.float32:
        return "f32"
    if t.dtype == torch.float64:
        return "f64"
    raise TypeError("nsol_amd kernels take float32 or float64, got %s" %
                    t.dtype)


def device():
    if not torch.cuda.is_available():
        raise RuntimeError(
            "nsol_amd needs a HIP device (torch.cuda.is_available() is False); "
            "there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def is_device_tensor(x):
    return isinstance(x, torch.Tensor) and x.is_cuda


def to_device(x, dtype=None):
    """NumPy array / torch tensor -> contiguous device tensor of `dtype`."""
    if isinstance(x, torch.Tensor):
        td = x.dtype if dtype is None else torch_dtype(dtype)
        if td not in (torch.float32, torch.float64):
            td = torch_dtype(_default_dtype)
        return x.to(device=device(), dtype=td).contiguous()
    a = np.asarray(x)
    if dtype is None:
        dtype = a.dtype.type if a.dtype in (np.float32, np.float64) \
            else np.float64
    a = np.ascontiguousarray(a, dtype=dtype)
    return torch.from_numpy(a).to(device())


def to_numpy(t, dtype=np.float64):
    return t.detach().cpu().numpy().astype(dtype, copy=False)


def empty_like(t, n=None):
    if n is None:
        return torch.empty_like(t)
    return torch.empty(int(n), dtype=t.dtype, device=t.device)


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream
