"""x = sum of 10 stored vectors (k_wcomb, the assembly of LSMR's solution) at 512^3
against the cap on the grid of element-wise kernels (knob max_grid_blocks)."""
import sys
import torch
sys.path.insert(0, ".")
from nsol_amd import ops, _lib
N = 512 ** 3
vecs = [torch.rand(N, device="cuda") for _ in range(10)]
co = [0.1 * (k + 1) for k in range(10)]
out = torch.empty(N, device="cuda")

def t(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

for rep in range(2):
    for blocks in (256, 512, 768, 1024, 1280, 1536):
        _lib.set_param("max_grid_blocks", blocks)
        print("max_grid_blocks=%d  wcomb %.4f  wcomb+clip %.4f" % (
            blocks, t(lambda: ops.lincomb_many(vecs, co, out=out)),
            t(lambda: ops.lincomb_many(vecs, co, out=out, bounds=(0.0, 1e30)))), flush=True)
_lib.set_param("max_grid_blocks", 2048)
