import sys
import torch
sys.path.insert(0, ".")
from nsol_amd import ops
n = 512
shape = (n, n, n); N = n ** 3
r = lambda m: torch.rand(m, device="cuda")
x, g, z, out = r(N), r(N), r(N), r(N)
slot = torch.zeros(1, dtype=torch.float64, device="cuda")
w = (1.0, 1.0, 1.0)
f = lambda: ops.tk1_lanczos(x, g, z, shape, w, 0.1, 0.5, -0.3, -0.2, out=out, result=slot)
for _ in range(10): f()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(40): f()
b.record(); torch.cuda.synchronize()
print("tk1_lanczos %.4f ms" % (a.elapsed_time(b) / 40))
