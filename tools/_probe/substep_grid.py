"""k_subspace_step / k_mdots / the Gram pass at 512^3 against the element-wise grid cap."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from nsol_amd import _lib
from nsol_amd.lbfgsb_device import DeviceBackend
n = 512 ** 3
be = DeviceBackend()
gen = torch.Generator(device="cuda").manual_seed(0)
r = lambda: torch.rand(n, device="cuda", generator=gen)
x, g, z, xcp = r(), r() - 0.5, r(), r()
free = (torch.rand(n, device="cuda", generator=gen) < 0.2).to(torch.int8)

def timed(fn, reps=8):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))

for c in (5, 10):
    ws = [r() for _ in range(c)]
    wy = [r() for _ in range(c)]
    coef = list(np.linspace(0.1, 1.0, c))
    for blocks in (2048, 1024, 512, 2048, 1024, 512):
        _lib.set_param("max_grid_blocks", blocks)
        a = timed(lambda: be.subspace_step(z, ws, wy, coef, coef, 0.7, free, xcp, x, g, 0.0, float("inf")))
        b = timed(lambda: be.dots(ws + wy, x, free))
        print("pairs=%d blocks=%d  subspace_step %.3f ms  mdots %.3f ms" % (c, blocks, a, b), flush=True)
    del ws, wy
_lib.set_param("max_grid_blocks", 2048)
