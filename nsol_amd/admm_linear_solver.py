"""ADMM for TV-regularised linear least squares on MI355X (drop-in for
nsol/admm_linear_solver.py:28-312):

    min_x 1/2 sum rho((A x - b)^2) + alpha ||B x - b_reg||_{2,1}

Each iteration (admm :202-218):
  1) x = TikhonovLinearSolver(alpha = rho, b_reg = v - w + b_reg, x0 = x)
  2) t = B x + w - b_reg;  v = isotropic shrink(t, alpha/rho);  w = t - v
Step 2 is ONE fused HIP kernel (grad + shrink + both updates + the next
right-hand side) when B is nsol_amd's gradient operator; otherwise it is
assembled from the generic kernels.
"""
import numpy as np

from . import ops
from .bridge import BridgedCallable
from .device import is_device_tensor
from .linear_solver import LinearSolver
from .symbolic import trace_operator
from ._accessors import add_accessors
from . import tikhonov_linear_solver as tk
from .definitions import EPS

# the fused outer step hands LSMR its lower right-hand side pre-multiplied by
# sqrt(rho) together with its norm (False: scale and norm passes per iteration)
USE_PRESCALED_RHS = True
# ... and, where the x-update is LSMR on the normal equations, the outer step forms the
# vector that solve starts from, A^T b + rho B^T(v - w + b_reg), in the same pass and never
# writes the right-hand side (ops.admm_vw_update_g: 36 B per voxel instead of 40 + 24)
USE_ONE_PASS_OUTER_STEP = True


class ADMMLinearSolver(LinearSolver):

    def __init__(self, A, A_adj, b, B, B_adj, x0, dimension, b_reg=0,
                 alpha=0.01, iter_max=10, minimizer="lsmr",
                 data_loss="linear", data_loss_scale=1, rho=0.5,
                 iterations=10, x_scale=1, verbose=0, dtype=None):
        LinearSolver.__init__(
            self, A=A, A_adj=A_adj, b=b, x0=x0, alpha=alpha, iter_max=iter_max,
            minimizer=minimizer, data_loss=data_loss,
            data_loss_scale=data_loss_scale, x_scale=x_scale, verbose=verbose,
            dtype=dtype)
        self._B = B
        self._B_adj = B_adj
        self._b_reg = self._scaled(b_reg)
        self._dimension = dimension
        self._rho = float(rho)
        self._iterations = iterations
        self._execution = None

    def get_execution(self):
        return self._execution

    # ------------------------------------------------------------------
    def _run(self):
        if self._observer is not None:
            self._observer.add_x(self.get_x())

        x = self._x0_device().clone()
        self._warm = None
        self._inner_log = []
        n = x.numel()
        B = BridgedCallable(self._B, self._dtype)
        desc = trace_operator(self._B, n)
        fused = desc is not None and desc[0] == "grad" and \
            desc[1].dimension == self._dimension and \
            len(desc[2]) == self._dimension and \
            int(np.prod(desc[2])) == n
        self._execution = "fused-outer" if fused else "generic-outer"

        scalar_c = not is_device_tensor(self._b_reg) and \
            np.ndim(self._b_reg) == 0
        c = None
        if not (scalar_c and float(self._b_reg) == 0.0):
            if scalar_c:
                import torch
                c = torch.full((B(x).numel(),), float(self._b_reg),
                               dtype=x.dtype, device=x.device)
            else:
                c = self._dev(self._b_reg)

        # v = B(x0) - b_reg ; w = 0                         (admm :171-172)
        v = B(x)
        if c is not None:
            v = ops.lincomb2(1.0, v, -1.0, c)
        import torch
        w = torch.zeros_like(v)
        # b_reg of the first x-update: v - w + b_reg        (admm :222)
        breg = ops.lincomb2(1.0, v, -1.0, w)
        if c is not None:
            breg = ops.lincomb2(1.0, breg, 1.0, c, out=breg)
        thr = self._alpha / self._rho
        # With the fused outer step and the fused LSMR inside, the step writes
        # the next right-hand side already multiplied by sqrt(rho) -- what the
        # augmented system wants (tikhonov :232-236) -- together with its sum of
        # squares, and ||b||^2 is taken once: no scaling pass and no norm passes
        # per ADMM iteration.
        prescale = fused and tk.USE_FUSED_LSMR and USE_PRESCALED_RHS and \
            self._minimizer == "lsmr" and self._data_loss == "linear" and \
            self._rho > EPS
        sa = float(np.sqrt(self._rho))
        hint = None
        rhs_read = self._minimizer in ("lsmr", "lsq_linear", "least_squares")
        b2 = ops.dot(self._dev(self._b), self._dev(self._b)) if prescale else None

        start = None
        for i in range(self._iterations):
            if self._verbose:
                print("ADMM iteration %d/%d" % (i + 1, self._iterations))
            x = self._solve_tikhonov_least_squares(x, breg, hint, start)
            # (v is read only through the next right-hand side: the fused step
            # does not store it)
            start = None
            if i + 1 == self._iterations:
                # (admm :208-216 after the last x-update changes v and w only, which
                # nobody reads any more: not done)
                self._x = x
                if self._observer is not None:
                    self._observer.add_x(self.get_x())
                break
            if fused and prescale and USE_ONE_PASS_OUTER_STEP:
                start = self._one_pass_outer_step(x, w, c, breg, desc, thr, sa)
            if start is not None:
                w, n2 = start.pop("w"), start.pop("n2")
                hint = (sa, b2, n2)        # (breg: written only if the solve asks)
            elif fused and prescale:
                n2 = ops.admm_vw_update(x, None, w, c, breg, desc[2], desc[1].w,
                                        thr, sa, want_norm=True)
                hint = (sa, b2, n2)        # breg holds sqrt(rho) * (v - w + c)
            elif fused:
                # (the robust-loss minimizers never read b_reg -- the reference's quirk,
                # tikhonov :201-208 -- so the next right-hand side is not written for
                # them: 28 instead of 40 B per voxel)
                ops.admm_vw_update(x, None, w, c, breg if rhs_read else None, desc[2],
                                   desc[1].w, thr, 1.0)
            else:
                Bx = B(x)
                t = ops.lincomb2(1.0, Bx, 1.0, w)
                if c is not None:
                    t = ops.lincomb2(1.0, t, -1.0, c, out=t)
                v = ops.vector_shrink(t, self._dimension, thr)
                w = ops.lincomb2(1.0, t, -1.0, v)
                breg = ops.lincomb2(1.0, v, -1.0, w)
                if c is not None:
                    breg = ops.lincomb2(1.0, breg, 1.0, c, out=breg)
            self._x = x
            if self._observer is not None:
                self._observer.add_x(self.get_x())
        self._x = x

    _warm = None
    _inner_log = ()

    def get_inner_log(self):
        """One record per inner solve of the last run (see _solve_tikhonov_least_squares)."""
        return list(self._inner_log)

    _w_alt = None
    _n2_fetch = None

    def _one_pass_outer_step(self, x, w, c, breg, desc, thr, sa):
        """v / w update and the next x-update's start vector in one pass; None when that
        kernel does not apply (nothing launched).  Returns what the inner solver needs
        ("g", "gg", "fill") plus the new w and the sum of squares of the right-hand
        side it stands for."""
        import torch
        if self._observer is not None:
            return None
        b = self._dev(self._b)
        atb = tk._adjoint_of_data(self._A_adj, BridgedCallable(self._A_adj, self._dtype),
                                  b)
        if atb.numel() != x.numel():
            return None
        if self._w_alt is None or self._w_alt.numel() != w.numel() or \
                self._w_alt.dtype != w.dtype or self._w_alt.data_ptr() == w.data_ptr():
            self._w_alt = torch.empty_like(w)
        g = torch.empty_like(x)
        sums = torch.empty(2, dtype=torch.float64, device=x.device)
        if not ops.admm_vw_update_g(x, w, self._w_alt, c, atb, g, desc[2], desc[1].w, thr,
                                    sa, 1.0, sa, sums):
            return None
        w_old, w_new = w, self._w_alt
        self._w_alt = w_old               # (next step's output; holds the old w until then)
        # the sum of squares of the right-hand side travels to the host on a side stream
        # and is asked for when the next solve's stopping tests first need it: the host
        # builds and enqueues that solve while this pass still runs (a read-back here
        # left the GPU idle for the ~0.4 ms that takes)
        if self._n2_fetch is None:
            self._n2_fetch = ops.ScalarFetch(x.device, 2)
        fetch = self._n2_fetch
        fetch.start(sums)
        memo = []

        def n2():
            if not memo:
                memo.append(float(fetch.wait()[0]))
            return memo[0]

        def fill():
            # the right-hand side itself after all: the two-kernel step from the old w
            ops.admm_vw_update(x, None, w_old.clone(), c, breg, desc[2], desc[1].w, thr,
                               sa)
        return {"g": g, "gg": sums[1:2], "fill": fill, "w": w_new, "n2": n2}

    def _solve_tikhonov_least_squares(self, x, b_reg, prescaled=None, start=None):
        # admm :220-237: data_loss_scale and bounds are NOT forwarded
        tikhonov = tk.TikhonovLinearSolver(
            A=self._A, A_adj=self._A_adj, B=self._B, B_adj=self._B_adj,
            b=self._dev(self._b), b_reg=b_reg, alpha=self._rho, x0=x,
            x_scale=1, iter_max=self._iter_max, data_loss=self._data_loss,
            minimizer=self._minimizer, verbose=self._verbose,
            dtype=self._dtype, _borrow=True)
        tikhonov._prescaled_b_reg = prescaled
        tikhonov._lsmr_start = start
        # (minimizer="L-BFGS-B": cost and gradient at the point the last solve
        # returned, see tikhonov_linear_solver.REUSE_OBJECTIVE_AT_X0)
        tikhonov._warm_key = ("admm", id(self), float(self._rho), self._data_loss,
                              self._minimizer)
        tikhonov._warm_start = self._warm
        tikhonov.run()
        self._warm = tikhonov._warm_result
        # what the inner solve decided (LSMR: SciPy's istop and the iterations taken;
        # L-BFGS-B: accepted iterations, evaluations, the task it ended on) -- two
        # precisions of one problem are compared through these
        info = getattr(tikhonov, "_minimize_info", None)
        self._inner_log.append(
            ("lsmr",) + tuple(getattr(tikhonov, "_lsmr_stop", (None, None)))
            if info is None else
            ("minimize", info.get("nit"), info.get("nfev"), info.get("task")))
        return tikhonov._x

    def _prox_g(self, t, tau, dimension):
        """Isotropic vector soft-threshold (admm :239-253); t: device tensor
        or NumPy array of `dimension` stacked blocks."""
        if is_device_tensor(t):
            return ops.vector_shrink(t.contiguous().view(-1), dimension,
                                     tau).view(t.shape)
        from .device import to_device, to_numpy
        arr = np.asarray(t, dtype=np.float64)
        out = ops.vector_shrink(to_device(arr.reshape(-1), np.float64),
                                dimension, tau)
        return to_numpy(out).reshape(arr.shape)

    def _get_cost_regularization_term(self, x):
        from .prior_measures import PriorMeasures
        return PriorMeasures.total_variation(x, self._B, self._dimension)


add_accessors(ADMMLinearSolver, ["rho", "iterations"])
add_accessors(ADMMLinearSolver, ["dimension"], setters=False)
