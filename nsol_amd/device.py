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


# Large device -> host copies go through two pinned staging buffers in chunks: the
# cast into the caller's dtype out of the staging buffer of chunk k overlaps the
# DMA of chunk k+1 (512^3 float32 -> float64 NumPy: 0.07-0.1 s instead of 0.19 s).
# Uploads stay plain: the runtime moves a pageable 512 MB array in ~11 ms here, and
# casts to the working dtype run on the device.
_CHUNK_BYTES = 64 << 20
_MIN_STAGED_BYTES = 32 << 20
_staging = {}


def _staging_buffers(nbytes):
    bufs = _staging.get("bufs")
    if bufs is None or bufs[0].numel() < nbytes:
        bufs = [torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
                for _ in range(2)]
        _staging["bufs"] = bufs
    return bufs


def _download(t, dtype):
    """Device tensor -> new NumPy array of `dtype` (cast on the way out of the
    staging buffer)."""
    if t.numel() * t.element_size() < _MIN_STAGED_BYTES:
        return t.detach().cpu().numpy().astype(dtype, copy=False)
    flat = t.detach().reshape(-1)
    res = np.empty(flat.numel(), dtype=dtype)
    per = max(1, _CHUNK_BYTES // flat.element_size())
    bufs = _staging_buffers(per * flat.element_size())
    pending = []           # (event, stage view, start, m)
    for k, start in enumerate(range(0, flat.numel(), per)):
        m = min(per, flat.numel() - start)
        if len(pending) == 2:
            ev, st, s0, m0 = pending.pop(0)
            ev.synchronize()
            res[s0:s0 + m0] = st.numpy()
        stage = bufs[k & 1][:m * flat.element_size()].view(flat.dtype)
        stage.copy_(flat[start:start + m], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        pending.append((ev, stage, start, m))
    for ev, st, s0, m0 in pending:
        ev.synchronize()
        res[s0:s0 + m0] = st.numpy()
    return res.reshape(tuple(t.shape))


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
    if a.dtype in (np.float32, np.float64) and \
            a.nbytes >= _MIN_STAGED_BYTES:
        # big arrays travel in their own dtype; the cast runs on the device
        t = torch.from_numpy(np.ascontiguousarray(a)).to(device())
        return t if t.dtype == torch_dtype(dtype) else t.to(torch_dtype(dtype))
    a = np.ascontiguousarray(a, dtype=dtype)
    return torch.from_numpy(a).to(device())


def to_numpy(t, dtype=np.float64):
    out = _download(t, dtype)
    # (the copy has synchronised the stream: settle the error words of the
    # persistent runs that produced what was just downloaded)
    import sys
    ops = sys.modules.get("nsol_amd.ops")
    if ops is not None and ops._pending_runs:
        if ops.settle_persist_runs(synchronize=False):
            out = _download(t, dtype)       # a run was repeated: fetch its result
    return out


def empty_like(t, n=None):
    if n is None:
        return torch.empty_like(t)
    return torch.empty(int(n), dtype=t.dtype, device=t.device)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr():
    """hipStream_t of torch's current stream on the current device (the raw query:
    a tenth of the cost of building a torch.cuda.Stream object, and an L-BFGS-B
    iteration asks a few dozen times)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream
