"""Chambolle-Pock primal-dual solver on MI355X (drop-in for
nsol/primal_dual_solver.py:26-403).

`run()` picks one of three execution forms:
  fused   B = grad, B_conj = grad_adj of nsol_amd.linear_operators, prox_g_conj
          in {prox_tv_conj, prox_huber_conj}, prox_f in {prox_ell1_denoising,
          prox_ell2_denoising} -- recognised THROUGH caller-side lambdas with
          a symbolic probe; nsol_pd_run_* enqueues the whole run: three
          iterations per pass over memory on large 3-D volumes (11 words of
          HBM traffic per voxel for all three; bit-identical to one
          iteration per launch), one single-pass kernel per iteration
          otherwise;
  device  any callables that work on torch HIP tensors (e.g.
          prox_linear_least_squares for deconvolution): the loop keeps all
          state in HBM and glues the callables with HIP axpy kernels; when
          only prox_f is foreign to the fused kernels, the regulariser side
          still runs fused (dual update in one pass, prox_f's argument in one
          pass: _run_native_dual);
  host    foreign NumPy-only callables: state stays in HBM, arguments are
          copied to the host for the callable only.
"""
import numpy as np

from . import ops
from .bridge import BridgedCallable
from .device import is_device_tensor
from .proximal_operators import scaled_data_on_device
from .solver import Solver
from ._accessors import add_accessors
from .symbolic import TauSym, trace_operator, trace_prox

# False: the "device" form glues every callable with separate axpy kernels even
# when the regulariser side is nsol_amd's own (the A/B reference of the tests)
USE_SEMI_FUSED = True
# 3-D volumes whose rows are not whole 16-byte vectors run with their arrays re-laid at
# a row pitch of whole vectors (nsol_pd_run_pitched_*), from this many voxels on
USE_ROW_PITCH = True
PITCH_MIN_VOXELS = 1 << 20


def step_schedule(alg_type, L2, lmbda, iterations):
    """Host-side step sizes; sigma[n], tau[n] are used inside iteration n and
    theta[n] in its over-relaxation (primal_dual_solver.py:222-253, 278-403).
    Unknown alg_type raises KeyError like the reference's dict lookup."""
    init = {"ALG2": _init_alg2, "ALG2_AHMOD": _init_alg2_ahmod,
            "ALG3": _init_alg3}[alg_type]
    tau, sigma, gamma = init(float(L2), lmbda)
    sig = np.empty(iterations)
    ta = np.empty(iterations)
    th = np.empty(iterations)
    for n in range(iterations):
        sig[n], ta[n] = sigma, tau
        if alg_type == "ALG3":
            theta = gamma            # constant steps; gamma carries theta
        else:
            theta = 1. / np.sqrt(1. + 2. * gamma * tau)
            tau = tau * theta
            sigma = sigma / theta
            if alg_type == "ALG2_AHMOD":
                theta = 0.
        th[n] = theta
    return sig, ta, th


def _init_alg2(L2, lmbda):
    tau0 = 1. / np.sqrt(L2)
    return tau0, 1. / (L2 * tau0), 0.35 * lmbda


def _init_alg2_ahmod(L2, lmbda):
    tau0 = 0.02
    return tau0, 4. / (L2 * tau0), 0.35 * lmbda


def _init_alg3(L2, lmbda, huber_alpha=0.05):
    mu = 2. * np.sqrt(lmbda * huber_alpha / L2)
    return mu / (2. * lmbda), mu / (2. * huber_alpha), 1. / (1. + mu)


class PrimalDualSolver(Solver):

    def __init__(self, prox_f, prox_g_conj, B, B_conj, L2, x0, alpha=0.01,
                 iterations=10, x_scale=1., verbose=0, alg_type="ALG2",
                 dtype=None):
        Solver.__init__(self, x0=x0, verbose=verbose, x_scale=x_scale,
                        dtype=dtype)
        self._prox_f = prox_f
        self._prox_g_conj = prox_g_conj
        self._B = B
        self._B_conj = B_conj
        self._L2 = float(L2)
        self._alpha = float(alpha)
        self._iterations = iterations
        self._alg_type = alg_type
        self._execution = None

    def get_execution(self):
        """'fused', 'device' or 'host' after run() (None before)."""
        return self._execution

    def print_statistics(self, fmt="%.3e"):
        pass

    # ------------------------------------------------------------------
    def _native_dual(self):
        """The regulariser side when it is nsol_amd's own: B = gradient, B_conj
        its adjoint, prox_g_conj = prox_tv_conj / prox_huber_conj.  Returns
        dict(shape, w, dim, flags, gamma) or None."""
        n = int(self._x0_host.size if self._x0_host is not None
                else self._x0_dev.numel())
        dB = trace_operator(self._B, n)
        if dB is None or dB[0] != "grad":
            return None
        gop, shape = dB[1], dB[2]
        if int(np.prod(shape)) != n or len(shape) != gop.dimension:
            return None
        dBt = trace_operator(self._B_conj, gop.dimension * n)
        if dBt is None or dBt[0] != "grad_adj":
            return None
        if tuple(dBt[1].w) != tuple(gop.w) or \
                dBt[1].dimension != gop.dimension or \
                tuple(dBt[2]) != tuple(gop._out_shape(shape)):
            return None
        dg = trace_prox(self._prox_g_conj, gop.dimension * n)
        if dg is None or dg[0] not in ("prox_tv_conj", "prox_huber_conj") \
                or not isinstance(dg[1], TauSym):
            return None
        return dict(shape=tuple(shape), w=gop.w, dim=gop.dimension, n=n,
                    flags=(ops.PD_REG_HUBER if dg[0] == "prox_huber_conj"
                           else ops.PD_REG_TV),
                    gamma=(dg[2] if dg[0] == "prox_huber_conj" else 0.05))

    def plan(self):
        """Recognise a fully native configuration.  Returns a dict for the
        fused kernel or None."""
        dual = self._native_dual()
        if dual is None:
            return None
        n = dual["n"]
        df = trace_prox(self._prox_f, n)
        if df is None or df[0] not in ("prox_ell1", "prox_ell2") \
                or not isinstance(df[3], TauSym):
            return None
        data = df[1]
        dsize = data.numel() if is_device_tensor(data) else np.size(data)
        if dsize != n:
            return None
        flags = dual["flags"]
        flags |= ops.PD_DATA_L1 if df[0] == "prox_ell1" else ops.PD_DATA_L2
        return dict(shape=dual["shape"], w=dual["w"], dim=dual["dim"],
                    flags=flags, gamma=dual["gamma"], data=data,
                    data_scale=df[2])

    def _run(self):
        if self._observer is not None:
            self._observer.add_x(self.get_x())
        lmbda = 1. / self._alpha
        sig, ta, th = step_schedule(self._alg_type, self._L2, lmbda,
                                    self._iterations)
        plan = self.plan()
        if plan is not None:
            self._execution = "fused"
            self._run_fused(plan, lmbda, sig, ta, th)
        else:
            self._run_generic(lmbda, sig, ta, th)

    # ------------------------------------------------------------------
    def _run_fused(self, plan, lmbda, sig, ta, th):
        import torch
        x = self._x0_device().clone()
        xbar = [x.clone(), torch.empty_like(x)]
        n = x.numel()
        p = [torch.empty(plan["dim"] * n, dtype=x.dtype, device=x.device)
             for _ in range(2)]
        bt = scaled_data_on_device(plan["data"], plan["data_scale"], x)
        pitch = ops.row_pitch(plan["shape"], x) if USE_ROW_PITCH and \
            n >= PITCH_MIN_VOXELS and self._iterations > 1 and \
            self._observer is None and not self._verbose else 0
        if pitch:
            # rows that are not whole 16-byte vectors (511^3, 181 x 217 x 181 ...): the
            # run's arrays hold them at a pitch of whole vectors -- aligned accesses
            # and whole stores instead of the ragged form's element-aligned ones (511^3:
            # the speed of 512^3 instead of +15...30 %); two re-layouts per run
            shape = plan["shape"]
            xq = ops.to_pitched(x, shape, pitch)
            np_ = xq.numel()
            xbq = [xq.clone(), torch.empty_like(xq)]
            pq = [torch.zeros(plan["dim"] * np_, dtype=x.dtype, device=x.device),
                  torch.empty(plan["dim"] * np_, dtype=x.dtype, device=x.device)]
            btq = ops.to_pitched(bt, shape, pitch)
            try:
                ops.pd_run(xbq[0], xbq[1], xq, btq, pq[0], pq[1], shape, plan["w"],
                           lmbda, sig, ta, th, True, plan["gamma"], plan["flags"],
                           x_alt=torch.zeros_like(xq), swap_ok=True, pitch=pitch)
                self._x = ops.from_pitched(xq, shape, pitch)
                return
            except ValueError:
                # the pitched entry declined (NSOL_EINVAL: the ragged-row form is
                # switched off, knob pd_rag): the contiguous arrays are untouched
                del xq, xbq, pq, btq
        if self._observer is None and not self._verbose:
            # scratch for the two-iterations-per-pass kernel (x ping-pong)
            x_alt = torch.empty_like(x) if self._iterations > 1 else None
            ops.pd_run(xbar[0], xbar[1], x, bt, p[0], p[1], plan["shape"],
                       plan["w"], lmbda, sig, ta, th, True, plan["gamma"],
                       plan["flags"], x_alt=x_alt, swap_ok=True)
            self._x = x
            return
        for i in range(self._iterations):      # observed / verbose: stepwise
            if self._verbose:
                print("Primal-Dual iteration %d/%d" % (i + 1,
                                                       self._iterations))
            k = i & 1
            hden = 1. + sig[i] * plan["gamma"] \
                if plan["flags"] & ops.PD_REG_HUBER else 1.
            ops.pd_fused_iter(xbar[k], xbar[1 - k], x, bt,
                              None if i == 0 else p[k], p[1 - k],
                              plan["shape"], plan["w"], sig[i], hden, ta[i],
                              ta[i] * lmbda, th[i], plan["flags"])
            self._x = x
            if self._observer is not None:
                self._observer.add_x(self.get_x())
        self._x = x

    # ------------------------------------------------------------------
    def _run_generic(self, lmbda, sig, ta, th):
        x = self._x0_device().clone()
        xbar = x.clone()
        pf = BridgedCallable(self._prox_f, self._dtype)
        dual = self._native_dual() if USE_SEMI_FUSED else None
        if dual is not None:
            self._run_native_dual(dual, pf, x, xbar, lmbda, sig, ta, th)
            return
        B = BridgedCallable(self._B, self._dtype)
        Bc = BridgedCallable(self._B_conj, self._dtype)
        pg = BridgedCallable(self._prox_g_conj, self._dtype)
        p = None
        for i in range(self._iterations):
            if self._verbose:
                print("Primal-Dual iteration %d/%d" % (i + 1,
                                                       self._iterations))
            g = B(xbar)
            # p + sigma * B(xbar); p = 0 before the first iteration
            q = ops.scale(g, sig[i]) if p is None else \
                ops.lincomb2(1.0, p, sig[i], g)
            p = pg(q, float(sig[i]))
            u = ops.lincomb2(1.0, x, -ta[i], Bc(p))
            x_new = pf(u, float(ta[i] * lmbda))
            # x_new + theta * (x_new - x)
            d = ops.lincomb2(1.0, x_new, -1.0, x)
            xbar = ops.lincomb2(1.0, x_new, th[i], d)
            x = x_new
            self._x = x
            if self._observer is not None:
                self._observer.add_x(self.get_x())
        self._x = x
        self._execution = "device" if all(
            c.on_device for c in (B, Bc, pg, pf)) else "host"

    def _run_native_dual(self, dual, pf, x, xbar, lmbda, sig, ta, th):
        """The loop of _run_generic when only prox_f is foreign to the fused
        kernels (e.g. prox_linear_least_squares: PD deconvolution, interface
        :257-280): the dual update in one pass (nsol_pd_dual_step_*: gradient,
        axpy and clamp, 28 bytes per voxel instead of 76), prox_f's argument
        x - tau K^T p in one pass (20 instead of 28) and the over-relaxation in
        one (12 instead of 24); the same values as the generic loop."""
        import torch
        shape, w = dual["shape"], dual["w"]
        p = torch.empty(dual["dim"] * x.numel(), dtype=x.dtype, device=x.device)
        huber = bool(dual["flags"] & ops.PD_REG_HUBER)
        for i in range(self._iterations):
            if self._verbose:
                print("Primal-Dual iteration %d/%d" % (i + 1,
                                                       self._iterations))
            hden = 1. + sig[i] * dual["gamma"] if huber else 1.
            ops.pd_dual_step(xbar, None if i == 0 else p, p, shape, w, sig[i],
                             hden)
            u = ops.grad_adj_axpy(p, x, ta[i], shape, w)
            x_new = pf(u, float(ta[i] * lmbda))
            xbar = ops.extrapolate(x_new, x, th[i], out=xbar)
            x = x_new
            self._x = x
            if self._observer is not None:
                self._observer.add_x(self.get_x())
        self._x = x
        self._execution = "device" if pf.on_device else "host"


add_accessors(PrimalDualSolver, ["alpha", "L2", "alg_type", "iterations"])
