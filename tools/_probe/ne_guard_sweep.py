#!/usr/bin/env python3
"""float32 error of LSMR as Lanczos on the normal equations against the float64 oracle
(SciPy's algorithm) over the regulariser's relative weight and the iteration count,
next to the Golub-Kahan form: sigma = 2 blur at 32^3, B = gradient / identity."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import nsol_amd.linear_operators as LO
import nsol_amd.tikhonov_linear_solver as tk
import nsol_amd.lsmr as L
from oracle import nsol_oracle as orc

n = 32
g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "cfg4.npz")))
y, ratio = g["y_32"], float(g["ratio_32"])
lo = LO.LinearOperators3D()
A, Aa = lo.get_gaussian_blurring_operators(np.diag([4.0] * 3))
grad, grad_adj = lo.get_gradient_operators()
X, Z = (n, n, n), (3 * n, n, n)
A_ = lambda x: A(x.reshape(*X)).flatten()
D_ = lambda x: grad(x.reshape(*X)).flatten()
Da_ = lambda x: grad_adj(x.reshape(*Z)).flatten()
I_ = lambda x: x.flatten()
Do, Dao, Ao, _ = orc.flat_operators(X, None, np.diag([4.0] * 3))
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
for bname in ("grad", "ident"):
    B, Ba = (D_, Da_) if bname == "grad" else (I_, I_)
    Bo, Bao = (Do, Dao) if bname == "grad" else (I_, I_)
    for wrel in (0.1, 0.05, 0.02, 0.01, 0.005):
        for iters in (10, 20, 32):
            w = wrel * ratio
            ref = orc.tikhonov(Ao, Ao, Bo, Bao, y, y, alpha=w, iter_max=iters,
                               x_scale=float(y.max()))
            out = {}
            for form in ("normal", "bidiag"):
                L.USE_NORMAL_EQUATIONS = form == "normal"
                L.NE_MIN_WEIGHT = {4: 0.0, 8: 0.0}
                L.LAST_NE_COND[0] = None
                s = tk.TikhonovLinearSolver(A=A_, A_adj=A_, B=B, B_adj=Ba, b=y, x0=y, alpha=w,
                                            x_scale=float(y.max()), iter_max=iters,
                                            dtype=np.float32)
                s.run()
                out[form] = rel(s.get_x(), ref)
                if form == "normal":
                    out["cond"] = L.LAST_NE_COND[0]
                    out["form"] = L.LAST_FORM[0]
            print(json.dumps({"B": bname, "weight_rel": wrel, "iters": iters, **out}), flush=True)
