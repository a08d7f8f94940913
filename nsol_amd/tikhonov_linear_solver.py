"""Tikhonov-regularised linear least squares on MI355X (drop-in for
nsol/tikhonov_linear_solver.py:30-280):

    min_x 1/2 sum rho((A x - b)^2) + alpha/2 ||B x - b_reg||^2

* minimizer "lsmr" (linear loss): GPU-resident LSMR on the augmented system
  [A; sqrt(alpha) B] x = [b; sqrt(alpha) b_reg] (tikhonov :146-158, :226-274),
  started from 0, result clipped to `bounds`.
* any other minimizer string goes to scipy.optimize.minimize exactly as the
  reference does (:197-220) -- including its quirk of ignoring b_reg there --
  with cost and gradient evaluated by HIP kernels; only the optimiser's own
  vector bookkeeping runs in SciPy on the host.
* "lsq_linear" / "least_squares" use SciPy's drivers over a LinearOperator whose
  matvec / rmatvec run on the GPU (:160-195).
"""
import numpy as np
import scipy.optimize
import scipy.sparse.linalg

from . import _caches, ops
from .bridge import BridgedCallable
from .definitions import EPS
from .device import is_device_tensor, to_device, to_numpy, torch_dtype
from .linear_solver import LinearSolver
from .lsmr import lsmr, lsmr_fused
from .solver import Solver
from .symbolic import trace_operator
from ._accessors import add_accessors


# set to False to force the generic (un-fused) LSMR vector kernels
USE_FUSED_LSMR = True
# minimizer="L-BFGS-B": True = GPU-resident driver (nsol_amd/lbfgsb.py, same
# algorithm and defaults as SciPy's, iterates agree to rounding); False = SciPy's
# host driver with GPU-evaluated cost / gradient
USE_DEVICE_LBFGSB = True
# robust-loss objective with B = gradient: 1/2||Bx||^2 and B_adj(Bx) from one
# pass over x (nsol_tk1_reg_cost_grad_*); False = grad, dot, grad_adj, lincomb2
USE_FUSED_TK1_REG = True
# ADMMLinearSolver with minimizer="L-BFGS-B" minimises the same objective in every
# outer iteration (the reference's cost and gradient ignore b_reg), each time from the
# point the last solve returned: True = that point's cost and gradient are handed to
# the next solve instead of being evaluated again (bit for bit the same values), and
# its projection onto the bounds (the identity on it) is not repeated
REUSE_OBJECTIVE_AT_X0 = True
# the device L-BFGS-B driver's g'd and projected-gradient norm of every new gradient
# from the kernel that forms it (nsol_tk1_reg_objective_*) instead of a pass and a
# read-back each (B = gradient only)
USE_OBJECTIVE_EXTRAS = True
# robust-loss objective with A = nsol_amd's blur and B = gradient: the data term
# rho'(r^2) r, 1/2 sum rho(r^2) as the epilogue of A x (nsol_corr3_wrap_loss_*) instead
# of a pass over a stored A x
USE_LOSS_EPILOGUE = True


# A^T b and |b|^2 for the (operator, data) pairs seen last: an outer loop (ADMM,
# primal-dual with prox_linear_least_squares) builds one solver per iteration around
# the same b.  nsol_amd/_caches.py says when a remembered value may be served.
_atb_cache = _caches.DataCache(2)
_bnorm_cache = _caches.DataCache(2)


def _adjoint_of_data(key_op, A_adj, b):
    # the operator enters by identity: the entry holds it (a callable, not a volume)
    hit = _atb_cache.lookup((b,), id(key_op))
    if hit is not None and hit[0] is key_op:
        return hit[1]
    val = A_adj(b)
    if val.untyped_storage().data_ptr() == b.untyped_storage().data_ptr():
        val = val.clone()                   # (an operator that hands back its argument)
    _atb_cache.store((b,), id(key_op), (key_op, val))
    return val


def _norm2_of_data(b):
    """|b|^2, kept like A^T b."""
    val = _bnorm_cache.lookup((b,))
    if val is None:
        val = _bnorm_cache.store((b,), None, ops.dot(b, b))
    return val


class TikhonovLinearSolver(LinearSolver):

    def __init__(self, A, A_adj, b, B, B_adj, x0, alpha=0.01, b_reg=0,
                 data_loss="linear", data_loss_scale=1, minimizer="lsmr",
                 iter_max=10, x_scale=1, verbose=0, bounds=(0, np.inf),
                 dtype=None, _borrow=False, _defer_scaling=False):
        LinearSolver.__init__(
            self, A=A, A_adj=A_adj, b=b, x0=x0, alpha=alpha, iter_max=iter_max,
            minimizer=minimizer, data_loss=data_loss,
            data_loss_scale=data_loss_scale, x_scale=x_scale, verbose=verbose,
            dtype=dtype, _borrow=_borrow)
        self._B = B
        self._B_adj = B_adj
        # _defer_scaling (a solver built for ONE solve around device data, e.g. by
        # prox_linear_least_squares): b_reg / x_scale is not formed here -- the LSMR
        # kernels take the factor as a coefficient -- but when somebody asks for it;
        # and the solution may come back in the caller's units (see _run_lsmr)
        self._fold_x_scale = bool(_defer_scaling) and float(self._x_scale) > 0 and \
            float(self._x_scale) != 1.0
        if self._fold_x_scale and is_device_tensor(b_reg) and self._borrowed(b_reg) is None:
            self._b_reg_val = None
            self._b_reg_lazy = (b_reg.to(torch_dtype(self._dtype)).contiguous().view(-1),
                                float(self._x_scale))
        else:
            self._b_reg = self._scaled(b_reg)          # tikhonov :91
        self._bounds = bounds
        # (sqrt(alpha), ||b||^2, ||sqrt(alpha) b_reg||^2) when b_reg already holds
        # sqrt(alpha) * b_reg (set by ADMMLinearSolver's fused outer step)
        self._prescaled_b_reg = None

    _x0_clip_pending = False
    _lsmr_start = None
    _b_reg_lazy = None
    _x_in_callers_units = False

    @property
    def _b_reg(self):
        if self._b_reg_lazy is not None:
            raw, xs = self._b_reg_lazy
            self._b_reg_lazy = None
            self._b_reg_val = ops.scale(raw, xs, divide=True)
        return self._b_reg_val

    @_b_reg.setter
    def _b_reg(self, v):
        self._b_reg_lazy = None
        self._b_reg_val = v

    def get_x_device(self):
        if self._x_in_callers_units and self._x is not None:
            return self._x.clone()
        return LinearSolver.get_x_device(self)

    def take_x_device(self):
        """The solution in the caller's units as a flat device tensor the caller may
        keep -- for a solver that is dropped after the call (no copy where the
        solution was assembled in those units already)."""
        if self._x_in_callers_units and self._x is not None:
            x, self._x = self._x, None
            self._x_in_callers_units = False
            return x
        return LinearSolver.get_x_device(self)

    def get_x(self):
        if self._x_in_callers_units and self._x is not None:
            return to_numpy(self._x)
        return LinearSolver.get_x(self)

    def _clip_x0(self):
        self._x0_clip_pending = False
        self._x0_dev = ops.clip(Solver._x0_device(self), self._bounds[0],
                                self._bounds[1])
        if self._x0_host is not None:
            self._x0_host = np.clip(self._x0_host, self._bounds[0],
                                    self._bounds[1])
        return self._x0_dev

    def _x0_device(self):
        if self._x0_clip_pending:
            return self._clip_x0()
        return Solver._x0_device(self)

    def get_x0(self):
        if self._x0_clip_pending:
            self._clip_x0()
        return Solver.get_x0(self)

    def get_b_reg(self):
        if is_device_tensor(self._b_reg):
            return to_numpy(ops.scale(self._b_reg, self._x_scale))
        return self._b_reg * self._x_scale

    # ------------------------------------------------------------------
    def _callables(self):
        dt = self._dtype
        return (BridgedCallable(self._A, dt), BridgedCallable(self._A_adj, dt),
                BridgedCallable(self._B, dt), BridgedCallable(self._B_adj, dt))

    def _run(self):
        if self._minimizer == "lsmr" and self._data_loss != "linear":
            raise ValueError(
                "lsmr solver cannot be used with non-linear data loss")
        elif self._minimizer == "lsq_linear" and self._data_loss != "linear":
            raise ValueError(
                "lsq_linear solver cannot be used with non-linear data loss")

        if self._observer is not None:
            self._observer.add_x(self.get_x())

        x0 = self._x0_device()
        lsmr_path = self._minimizer == "lsmr" and self._data_loss == "linear"
        if self._bounds is not None:                       # tikhonov :142-143
            if lsmr_path and self._observer is None and self._x0_host is None:
                # LSMR starts from zero (SciPy's default): x0 only gives the
                # shape, so its projection waits until somebody asks for x0
                self._x0_clip_pending = True
            elif self._warm_start_applies(x0):
                pass                # (a point this minimizer returned: inside the bounds)
            else:
                x0 = self._clip_x0()

        if lsmr_path:
            self._x = self._run_lsmr(x0)        # (projected onto the bounds)
        elif self._minimizer in ("lsq_linear", "least_squares"):
            self._x = self._run_scipy_least_squares(x0)
        else:
            self._x = self._run_minimize(x0)

        if self._observer is not None:
            self._observer.add_x(self.get_x())

    # ------------------------------------------------------------------
    def _augmented(self, x0):
        """Block form of tikhonov :226-274: (matvec, rmatvec, rhs blocks)."""
        A, A_adj, B, B_adj = self._callables()
        b = self._dev(self._b)
        if self._alpha > EPS:
            sa = float(np.sqrt(self._alpha))
            if is_device_tensor(self._b_reg) or np.ndim(self._b_reg) > 0:
                lower = ops.scale(self._dev(self._b_reg), sa)
            else:
                # scalar b_reg (default 0): broadcast over the rows of B
                nrows = B(x0).numel()
                import torch
                lower = torch.full((nrows,), sa * float(self._b_reg),
                                   dtype=x0.dtype, device=x0.device)

            def matvec(v):
                return [A(v), ops.scale(B(v), sa)]

            def rmatvec(u):
                return ops.lincomb2(1.0, A_adj(u[0]), sa, B_adj(u[1]))
            return matvec, rmatvec, [b.clone(), lower]

        def matvec(v):
            return [A(v)]

        def rmatvec(u):
            return A_adj(u[0])
        return matvec, rmatvec, [b.clone()]

    def _run_lsmr(self, x0):
        fused = self._fused_lsmr_setup(x0) if USE_FUSED_LSMR else None
        pre = self._prescaled_b_reg
        # (set by ADMMLinearSolver's one-pass outer step: the vector LSMR starts from is
        # formed already and b_reg is not -- "fill" writes it should the solve need it)
        start = self._lsmr_start if fused is not None and pre is not None else None
        if start is None and self._lsmr_start is not None:
            self._lsmr_start["fill"]()
        if fused is not None:
            # (b is handed over as it is: the bidiagonalisation takes a copy to work
            # in, the normal-equations form only reads it -- and A^T b, the same in
            # every solve of an outer loop around one b, is kept)
            b_top = fused[2]
            # (x_scale folded into the coefficients that assemble x, where the solve
            # gets that far: out_scale[1] says whether it did)
            out_scale = [float(self._x_scale), False] if self._fold_x_scale else None
            x, istop, itn = lsmr_fused(*fused, x_like=x0, maxiter=self._iter_max,
                                 out_scale=out_scale,
                                 A_axpby=self._blur_epilogue(x0.numel()),
                                 normb2=None if pre is None else (
                                     (lambda: pre[1] + pre[2]()) if callable(pre[2])
                                     else pre[1] + pre[2]),
                                 own_b=False,
                                 atb=lambda: _adjoint_of_data(self._A_adj, fused[1],
                                                              b_top),
                                 top_norm2=lambda: _norm2_of_data(b_top),
                                 x_bounds=self._bounds,
                                 b_bot_scale=self._lower_scale,
                                 g0=None if start is None else (start["g"], start["gg"]),
                                 b_bot_fill=None if start is None else start["fill"])
            self._lsmr_stop = (istop, itn)     # (SciPy's istop, iterations taken)
            self._x_in_callers_units = bool(out_scale and out_scale[1])
            return x
        if pre is not None:            # (not expected: undo the pre-multiplication)
            self._b_reg = ops.scale(self._dev(self._b_reg), 1.0 / pre[0])
            self._prescaled_b_reg = None
        matvec, rmatvec, rhs = self._augmented(x0)
        x, istop, itn = lsmr(matvec, rmatvec, rhs, x0, self._iter_max)
        self._lsmr_stop = (istop, itn)
        if self._bounds is not None:                       # tikhonov :142-143
            x = ops.clip(x, self._bounds[0], self._bounds[1], out=x)
        return x

    def _blur_epilogue(self, n):
        """A_axpby(v, io, ca, cb) when A is nsol_amd's convolution operator seen
        through the caller's reshape / flatten lambda (its one-pass blur can form
        the top block of LSMR's u update itself), else None."""
        d = trace_operator(self._A, n)
        if d is None or d[0] != "conv" or int(np.prod(d[2])) != n:
            return None
        op, shape = d[1], tuple(d[2])
        if not hasattr(op, "apply_axpby"):
            return None
        def epilogue(v, io, ca, cb, result=None):
            return op.apply_axpby(v, io, shape, ca, cb, result=result)
        # (for lsmr_normal: the blur that also takes sum |grad v|^2 of its input,
        # where the regulariser's gradient runs over the same 3-D grid)
        epilogue.shape = shape
        epilogue.norms = lambda v, out, w, result: op.apply_norms(v, out, shape, w,
                                                                  result)
        # (for lsmr_normal: both halves of a Lanczos step inside the blur -- needs
        # A_adj to be this very blur, which it is for symmetric taps)
        epilogue.lanczos = None
        da = trace_operator(self._A_adj, n)
        if da is not None and da[0] == "conv" and tuple(da[2]) == shape and \
                hasattr(op, "lanczos_halves") and \
                (da[1] is op or getattr(da[1], "same_blur_as", lambda o: False)(op)):
            epilogue.lanczos = op.lanczos_halves(shape)
        return epilogue

    def _fused_lsmr_setup(self, x0):
        """Arguments for lsmr_fused when the regulariser operator is
        recognised (gradient / identity / absent); None otherwise."""
        n = x0.numel()
        A, A_adj, _, _ = self._callables()
        b = self._dev(self._b)
        if b.numel() != n:
            return None
        flat_ok = ops.flat_geometry(n) is not None
        self._lower_scale = 1.0
        if not (self._alpha > EPS):
            if not flat_ok:
                return None
            return (A, A_adj, b, None, ops.B_NONE, (n,),
                    (1.0, 1.0, 1.0), 0.0)
        dB = trace_operator(self._B, n)
        if dB is None:
            return None
        if dB[0] == "identity":
            if not flat_ok:
                return None
            bmode, shape, w, rows = ops.B_IDENTITY, (n,), (1., 1., 1.), n
            dBt = trace_operator(self._B_adj, n)
            if dBt is None or dBt[0] != "identity":
                return None
        elif dB[0] == "grad":
            gop, shape = dB[1], tuple(dB[2])
            if len(shape) != gop.dimension or int(np.prod(shape)) != n:
                return None
            dBt = trace_operator(self._B_adj, gop.dimension * n)
            if dBt is None or dBt[0] != "grad_adj" or \
                    tuple(dBt[1].w) != tuple(gop.w) or \
                    dBt[1].dimension != gop.dimension:
                return None
            bmode, w, rows = ops.B_GRAD, gop.w, gop.dimension * n
        else:
            return None
        sa = float(np.sqrt(self._alpha))
        self._lower_scale = 1.0
        lazy = self._b_reg_lazy
        if self._prescaled_b_reg is None and lazy is not None and lazy[0].numel() == rows:
            # (b_reg as the caller gave it; 1 / x_scale rides with sqrt(alpha))
            lower = lazy[0]
            self._lower_scale = sa / lazy[1]
        elif self._prescaled_b_reg is not None and \
                is_device_tensor(self._b_reg) and self._b_reg.numel() == rows:
            # already sqrt(alpha) * b_reg; consumed by LSMR (it becomes u's lower
            # block), which is fine: the caller rewrites it before the next solve
            lower = self._dev(self._b_reg)
        elif is_device_tensor(self._b_reg) or np.ndim(self._b_reg) > 0:
            if self._prescaled_b_reg is not None:
                return None
            # (handed over as it is with its factor: only the bidiagonalisation needs
            # sqrt(alpha) * b_reg as an array of its own)
            lower = self._dev(self._b_reg)
            self._lower_scale = sa
            if lower.numel() != rows:
                return None
        else:
            if self._prescaled_b_reg is not None:
                return None
            import torch
            lower = torch.full((rows,), sa * float(self._b_reg),
                               dtype=x0.dtype, device=x0.device)
        return (A, A_adj, b, lower, bmode, shape, w, sa)

    # ------------------------------------------------------------------
    def _host_linear_operator(self, x0):
        matvec, rmatvec, rhs = self._augmented(x0)
        sizes = [r.numel() for r in rhs]
        dt = self._dtype

        def mv(v):
            parts = matvec(to_device(np.asarray(v, dtype=np.float64)
                                     .reshape(-1), dt))
            return np.concatenate([to_numpy(p) for p in parts])

        def rmv(u):
            u = np.asarray(u, dtype=np.float64).reshape(-1)
            parts, o = [], 0
            for s in sizes:
                parts.append(to_device(u[o:o + s], dt))
                o += s
            return to_numpy(rmatvec(parts))
        op = scipy.sparse.linalg.LinearOperator(
            shape=(sum(sizes), x0.numel()), matvec=mv, rmatvec=rmv,
            dtype=np.float64)
        return op, np.concatenate([to_numpy(r) for r in rhs])

    def _run_scipy_least_squares(self, x0):
        op, rhs = self._host_linear_operator(x0)
        if self._minimizer == "lsq_linear":
            x = scipy.optimize.lsq_linear(
                op, rhs, max_iter=self._iter_max, lsq_solver='lsmr',
                lsmr_tol='auto', bounds=self._bounds,
                verbose=2 * self._verbose).x
        else:
            x = scipy.optimize.least_squares(
                fun=lambda x: op * x - rhs, jac=lambda x: op,
                jac_sparsity=lambda x: op, x0=to_numpy(x0), tr_solver='lsmr',
                bounds=self._bounds, loss=self._data_loss,
                f_scale=self._data_loss_scale, max_nfev=self._iter_max,
                verbose=2 * self._verbose).x
        return to_device(x, self._dtype)

    # ------------------------------------------------------------------
    def _device_objective(self):
        """cost(x), gradient(x) of tikhonov :201-208 on device vectors."""
        A, A_adj, B, B_adj = self._callables()
        b = self._dev(self._b)
        use_reg = self._alpha > EPS
        alpha = self._alpha
        loss, fscale = self._data_loss, self._data_loss_scale

        native = self._native_gradient(b.numel()) \
            if use_reg and USE_FUSED_TK1_REG else None

        slots = []
        # A = nsol_amd's blur seen through the caller's lambda: the data term as the
        # epilogue of A x (no A x in memory, no pass of its own)
        blur_loss = None
        if native is not None and USE_LOSS_EPILOGUE:
            d = trace_operator(self._A, b.numel())
            if d is not None and d[0] == "conv" and int(np.prod(d[2])) == b.numel() \
                    and hasattr(d[1], "apply_loss"):
                blur_loss = (d[1], tuple(d[2]))

        def fun_and_grad(x, extras=None):
            if blur_loss is not None:
                import torch
                if not slots:
                    slots.append(torch.empty(5, dtype=torch.float64, device=x.device))
                g = blur_loss[0].apply_loss(x, b, blur_loss[1], loss, fscale,
                                            slots[0][0:1])
                if g is not None:
                    return finish(x, g, extras)
            r = A(x)
            # in place unless A handed x itself back (an identity operator)
            own = r.untyped_storage().data_ptr() != x.untyped_storage().data_ptr()
            if native is not None:
                # B = gradient, B_adj its adjoint: 1/2||Bx||^2 and B_adj(Bx)
                # from one pass over x (same values as the branch below).  The
                # sums wait in device slots until all four kernels are
                # enqueued: one read-back instead of two, behind them.  With
                # `extras` = (d, lo, hi) the last kernel also leaves g'd and the
                # largest projected component of g there (what L-BFGS-B asks of
                # every new gradient: no pass of their own, no further read-back)
                import torch
                if not slots:
                    slots.append(torch.empty(5, dtype=torch.float64,
                                             device=r.device))
                _, g = ops.loss_cost_grad(r, loss, fscale,
                                          out=r if own else None, minus=b,
                                          result=slots[0][0:1])
                return finish(x, g, extras)
            cost, g = ops.loss_cost_grad(r, loss, fscale, out=r if own else None,
                                         minus=b)
            grad = A_adj(g)
            if use_reg:
                Bx = B(x)
                # reference quirk kept: 1/2||Bx||^2, b_reg is ignored here
                cost = cost + alpha * (0.5 * ops.dot(Bx, Bx))
                grad = ops.lincomb2(1.0, grad, alpha, B_adj(Bx))
            return cost, grad

        def finish(x, g, extras):
            """The rest of the native evaluation from g = rho'(r^2) r (its cost waits in
            slots[0][0])."""
            grad = A_adj(g)
            shape, w = native
            if extras is not None:
                d, lo, hi, gold = extras
                ydiff = ops.empty_like(grad) if gold is not None else None
                ops.tk1_reg_objective(x, grad, d, shape, w, alpha, lo, hi,
                                      out=grad, result=slots[0][1:5], gold=gold,
                                      ydiff=ydiff)
                sums = slots[0].cpu()
                return (float(sums[0]) + alpha * (0.5 * float(sums[1])), grad,
                        float(sums[2]) if d is not None else None,
                        float(sums[3]), ydiff,
                        float(sums[4]) if gold is not None else None)
            _, grad = ops.tk1_reg_cost_grad(x, grad, shape, w, alpha,
                                            out=grad, result=slots[0][1:2])
            sums = slots[0][:2].cpu()
            return float(sums[0]) + alpha * (0.5 * float(sums[1])), grad
        if native is not None and USE_OBJECTIVE_EXTRAS:
            # (f, g, g'd, |proj g|_inf, g - gold, its squared norm) from the same kernels:
            # lbfgsb.minimize's protocol
            fun_and_grad.with_extras = lambda x, d, lo, hi, gold=None: \
                fun_and_grad(x, (d, lo, hi, gold))
        return fun_and_grad

    _warm_start = None              # set by the outer solver (see _run_minimize)
    _warm_result = None

    def _warm_start_applies(self, x0):
        warm = self._warm_start
        return REUSE_OBJECTIVE_AT_X0 and warm is not None and \
            self._minimizer == "L-BFGS-B" and USE_DEVICE_LBFGSB and \
            is_device_tensor(x0) and warm["x"].data_ptr() == x0.data_ptr() and \
            warm["x"].numel() == x0.numel() and warm["x"].dtype == x0.dtype and \
            warm["x"].untyped_storage() is x0.untyped_storage() and \
            int(x0._version) == warm["version"] and \
            self._bounds is not None and \
            warm["bounds"] == (float(self._bounds[0]), float(self._bounds[1])) and \
            self._warm_key is not None and warm["key"] == self._objective_key()

    # named by the outer solver that builds one solver per iteration around the same
    # operators, data, weight and loss (its own identity): what _device_objective's
    # value depends on besides x (b_reg does not enter it: the reference's quirk)
    _warm_key = None

    def _objective_key(self):
        return self._warm_key

    def _native_gradient(self, n):
        """(shape, inverse spacings) when B / B_adj are nsol_amd's gradient
        and its adjoint on an n-voxel volume; None otherwise."""
        dB = trace_operator(self._B, n)
        if dB is None or dB[0] != "grad":
            return None
        gop, shape = dB[1], tuple(dB[2])
        if len(shape) != gop.dimension or int(np.prod(shape)) != n:
            return None
        dBt = trace_operator(self._B_adj, gop.dimension * n)
        if dBt is None or dBt[0] != "grad_adj" or \
                tuple(dBt[1].w) != tuple(gop.w) or \
                dBt[1].dimension != gop.dimension:
            return None
        return shape, gop.w

    def _run_minimize(self, x0):
        if self._minimizer == "L-BFGS-B" and USE_DEVICE_LBFGSB:
            from . import lbfgsb
            from .lbfgsb_device import DeviceBackend
            lo, hi = self._bounds
            # An outer solver that minimises the SAME objective again from the point the
            # last solve returned (ADMMLinearSolver with this minimizer: the reference's
            # cost ignores b_reg, tikhonov :201-208) hands over that point's f and g:
            # the first evaluation would recompute them bit for bit
            start = None
            if self._warm_start_applies(x0):
                start = (self._warm_start["f"], self._warm_start["g"],
                         self._warm_start.get("pg"))
            x, info = lbfgsb.minimize(self._device_objective(), x0, float(lo),
                                      float(hi), DeviceBackend(),
                                      maxiter=self._iter_max, start=start)
            self._minimize_info = info
            self._warm_result = None
            if info.get("jac") is not None:
                # (x as it is now: a write to it that torch or nsol_amd.ops can see
                # -- its version counter -- withdraws the hand-over)
                self._warm_result = {"x": x, "version": int(x._version),
                                     "f": info["fun"], "g": info["jac"],
                                     "pg": info.get("pg"),
                                     "bounds": (float(lo), float(hi)),
                                     "key": self._objective_key(),
                                     "reused": start is not None}
            return x
        A, A_adj, B, B_adj = self._callables()
        b = self._dev(self._b)
        dt = self._dtype
        use_reg = self._alpha > EPS
        alpha = self._alpha
        loss, fscale = self._data_loss, self._data_loss_scale

        def fun_and_jac(xh):
            x = to_device(np.asarray(xh, dtype=np.float64).reshape(-1), dt)
            r = ops.lincomb2(1.0, A(x), -1.0, b)
            cost, g = ops.loss_cost_grad(r, loss, fscale, out=r)
            grad = A_adj(g)
            if use_reg:
                Bx = B(x)
                # reference quirk kept: 1/2||Bx||^2, b_reg is ignored here
                cost = cost + alpha * (0.5 * ops.dot(Bx, Bx))
                grad = ops.lincomb2(1.0, grad, alpha, B_adj(Bx))
            return cost, to_numpy(grad)

        n = x0.numel()
        lo, hi = self._bounds
        bounds = scipy.optimize.Bounds(np.full(n, lo, dtype=np.float64),
                                       np.full(n, hi, dtype=np.float64))
        res = scipy.optimize.minimize(
            method=self._minimizer, fun=fun_and_jac, jac=True,
            x0=to_numpy(x0), bounds=bounds,
            options={'maxiter': self._iter_max, 'disp': self._verbose})
        return to_device(res.x, dt)

    def _get_cost_regularization_term(self, x):
        Bx = BridgedCallable(self._B, self._dtype)(self._dev(x))
        return 0.5 * ops.dot(Bx, Bx)

    def _get_gradient_cost_regularization_term(self, x):
        B = BridgedCallable(self._B, self._dtype)
        Ba = BridgedCallable(self._B_adj, self._dtype)
        return Ba(B(self._dev(x)))


add_accessors(TikhonovLinearSolver, ["B", "B_adj"], setters=False)
