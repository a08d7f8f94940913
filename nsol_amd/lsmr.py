"""LSMR (Fong & Saunders, SIAM J. Sci. Comput. 2011) with all vectors resident
in HBM.  Replaces the call
    scipy.sparse.linalg.lsmr(A, b, maxiter=iter_max, atol=0, btol=0)
of tikhonov_linear_solver.py:149-154 (SciPy 1.15 semantics: damp = 0,
conlim = 1e8, start from x = 0).

The operator is given in block form so that the augmented system
[A; sqrt(alpha) B] never needs a concatenated copy:
    matvec(v)      -> list of device vectors  (one per row block)
    rmatvec(parts) -> device vector
Golub-Kahan vectors are updated by HIP axpy kernels, norms are deterministic
float64 reductions on the GPU; the plane rotations are host scalars.
"""
import math

import numpy as np

from . import ops


def _sym_ortho(a, b):
    """Stable Givens rotation: returns (c, s, r) with c*a + s*b = r."""
    if b == 0:
        return float(np.sign(a)), 0.0, abs(a)
    if a == 0:
        return 0.0, float(np.sign(b)), abs(b)
    if abs(b) > abs(a):
        t = a / b
        s = float(np.sign(b)) / math.sqrt(1 + t * t)
        return s * t, s, b / s
    t = b / a
    c = float(np.sign(a)) / math.sqrt(1 + t * t)
    return c, c * t, a / c


def _norm(parts):
    return math.sqrt(sum(ops.dot(p, p) for p in parts))


def lsmr(matvec, rmatvec, b_parts, x_like, maxiter, atol=0.0, btol=0.0,
         conlim=1e8):
    """Minimise ||A x - b||_2.  b_parts: list of device vectors (consumed:
    they become the u blocks).  x_like: device vector shaped like x.
    Returns (x, istop, itn)."""
    import torch
    u = list(b_parts)
    normb = _norm(u)
    x = torch.zeros_like(x_like)
    beta = normb
    if beta > 0:
        for k in range(len(u)):
            ops.scale(u[k], 1.0 / beta, out=u[k])
        v = rmatvec(u)
        alpha = _norm([v])
    else:
        v = torch.zeros_like(x_like)
        alpha = 0.0
    if alpha > 0:
        ops.scale(v, 1.0 / alpha, out=v)

    itn = 0
    zetabar = alpha * beta
    alphabar = alpha
    rho = rhobar = cbar = 1.0
    sbar = 0.0
    h = v.clone()
    hbar = torch.zeros_like(x_like)
    betadd, betad = beta, 0.0
    rhodold = 1.0
    tautildeold = thetatilde = zeta = d = 0.0
    normA2 = alpha * alpha
    maxrbar, minrbar = 0.0, 1e100
    istop = 0
    ctol = 1.0 / conlim if conlim > 0 else 0.0
    if alpha * beta == 0 or normb == 0:
        return x, istop, itn

    while itn < maxiter:
        itn += 1
        # bidiagonalisation:  beta u = A v - alpha u ;  alpha v = A^T u - beta v
        Av = matvec(v)
        for k in range(len(u)):
            ops.lincomb2(-alpha, u[k], 1.0, Av[k], out=u[k])
        del Av
        beta = _norm(u)
        if beta > 0:
            for k in range(len(u)):
                ops.scale(u[k], 1.0 / beta, out=u[k])
            ops.lincomb2(-beta, v, 1.0, rmatvec(u), out=v)
            alpha = _norm([v])
            if alpha > 0:
                ops.scale(v, 1.0 / alpha, out=v)

        chat, shat, alphahat = _sym_ortho(alphabar, 0.0)
        rhoold = rho
        c, s, rho = _sym_ortho(alphahat, beta)
        thetanew = s * alpha
        alphabar = c * alpha
        rhobarold, zetaold = rhobar, zeta
        thetabar = sbar * rho
        rhotemp = cbar * rho
        cbar, sbar, rhobar = _sym_ortho(cbar * rho, thetanew)
        zeta = cbar * zetabar
        zetabar = -sbar * zetabar

        ops.lincomb2(-(thetabar * rho / (rhoold * rhobarold)), hbar, 1.0, h,
                     out=hbar)
        ops.lincomb2(1.0, x, zeta / (rho * rhobar), hbar, out=x)
        ops.lincomb2(-(thetanew / rho), h, 1.0, v, out=h)

        betaacute = chat * betadd
        betacheck = -shat * betadd
        betahat = c * betaacute
        betadd = -s * betaacute
        thetatildeold = thetatilde
        ctildeold, stildeold, rhotildeold = _sym_ortho(rhodold, thetabar)
        thetatilde = stildeold * rhobar
        rhodold = ctildeold * rhobar
        betad = -stildeold * betad + ctildeold * betahat
        tautildeold = (zetaold - thetatildeold * tautildeold) / rhotildeold
        taud = (zeta - thetatilde * tautildeold) / rhodold
        d = d + betacheck * betacheck
        normr = math.sqrt(d + (betad - taud) ** 2 + betadd * betadd)
        normA2 = normA2 + beta * beta
        normA = math.sqrt(normA2)
        normA2 = normA2 + alpha * alpha
        maxrbar = max(maxrbar, rhobarold)
        if itn > 1:
            minrbar = min(minrbar, rhobarold)
        condA = max(maxrbar, rhotemp) / min(minrbar, rhotemp)

        normar = abs(zetabar)
        normx = _norm([x])
        test1 = normr / normb
        test2 = normar / (normA * normr) if (normA * normr) != 0 else np.inf
        test3 = 1.0 / condA
        t1 = test1 / (1 + normA * normx / normb)
        rtol = btol + atol * normA * normx / normb
        if itn >= maxiter:
            istop = 7
        if 1 + test3 <= 1:
            istop = 6
        if 1 + test2 <= 1:
            istop = 5
        if 1 + t1 <= 1:
            istop = 4
        if test3 <= ctol:
            istop = 3
        if test2 <= atol:
            istop = 2
        if test1 <= rtol:
            istop = 1
        if istop > 0:
            break
    return x, istop, itn


# lsmr_fused keeps every Golub-Kahan vector v_k (the update writes v_{k+1} into a
# buffer of its own) and assembles x = sum_k a_k v_k in ONE pass at the end: h,
# hbar and x are linear combinations of the v_k whose coefficients follow SciPy's
# recurrences (lsmr.py:352-364) on the host, so the 28 bytes per element and
# iteration of that update (a quarter of an iteration's traffic) are not moved at
# all.  Held to DEFER_X_BYTES of stored vectors; False: the three vectors are
# carried through every iteration (nsol_lsmr_hx_update_*; the A/B reference).
DEFER_X = True
DEFER_X_BYTES = 48 << 30
_MAX_COMBINED = 40           # vectors one nsol_lb_wcomb_* launch combines


class SolutionCoefficients(object):
    """h, hbar and x of SciPy's loop (lsmr.py:352-364) as coefficient vectors over
    the normalised Golub-Kahan vectors v_1 .. v_K:
        hbar <- h + c_hbar * hbar ;  x <- x + c_x * hbar ;  h <- v_new + c_h * h
    with h_1 = v_1, hbar_0 = x_0 = 0.  `newest` is the index of v_new among the
    kept vectors (the last one; unchanged when the bidiagonalisation broke down
    and no new vector was formed)."""

    def __init__(self, capacity):
        self.h = np.zeros(capacity)
        self.hbar = np.zeros(capacity)
        self.x = np.zeros(capacity)
        self.h[0] = 1.0

    def step(self, c_hbar, c_x, c_h, newest):
        self.hbar = self.h + c_hbar * self.hbar
        self.x = self.x + c_x * self.hbar
        self.h = c_h * self.h
        self.h[newest] += 1.0

    def normx2(self):
        """||x||^2 for orthonormal v_k."""
        return float(np.dot(self.x, self.x))


# LSMR is MINRES on the normal equations M x = A^T b with M = A^T A + sa^2 B^T B in
# exact arithmetic (Fong & Saunders 2011, section 2): same Krylov space, same
# minimised quantity ||A^T r||, same iterates x_k.  Golub-Kahan keeps the rows of the
# augmented operator as vectors -- with B = gradient a field of three components per
# voxel that is read and written twice per iteration (36 of the 72 bytes).  Lanczos
# on M only ever holds vectors the size of x:
#     t = A y_j ,  y' = A^T t + sa^2 B^T B y_j ,  alfa = (||t||^2 + sa^2 ||B y_j||^2) / beta_j^2
#     y_{j+1} = y'/beta_j - (alfa/beta_j) y_j - (beta_j/beta_{j-1}) y_{j-1} ,  beta_{j+1} = ||y_{j+1}||
# (y_j unnormalised Lanczos vectors, kept; 48 bytes per voxel and iteration with
# B = gradient, 36 with the identity), the Paige-Saunders rotations run on the host
# and x = sum_j a_j y_j is assembled once, as above.  Squaring the operator costs
# accuracy where M is ill-conditioned.  Measured against SciPy's lsmr (float64) on
# config 4's operator (sigma = 2 blur, 10 iterations; float64 / float32 vectors):
#     gradient, weight 0.5: 1e-15 / 6e-8     identity, weight 4:   7e-16 / 7e-8
#     gradient, weight 0.1: 5e-15 / 1.4e-7   identity, weight 0.5: 4e-15 / 1.6e-7
#     gradient, weight .05: 1e-14 / 1.3e-6   identity, weight .05: 4e-14 / 1.1e-6
#     no regulariser: 8e-14 / 1.9e-6 (30 iterations: 5e-6 / 4e-2)
# (Golub-Kahan in float32: 1.3e-7 throughout.)  Hence the form is taken only with a
# regulariser of weight sa^2 >= 0.1 in float32 (1e-2 in float64) and for at most
# NE_MAX_ITER iterations -- BASELINE config 4 (rho = 0.1), the ADMM goldens
# (rho = 0.5) and primal-dual deconvolution (weight 1 / tau >= 1) qualify; everything
# else runs the bidiagonalisation.
# The weight is taken RELATIVE to the operator's own scale (a caller's blur kernel need
# not sum to one): against ||A v_1||^2 of the first, data-like Lanczos vector, which the
# first step measures anyway (1 for a normalised blur; a run that fails the test has
# spent three kernels and continues as a bidiagonalisation).  Large weights are
# harmless (gradient, weight 1 / 4 / 16: 9e-8 / 1e-7 / 1.5e-7 in float32).  The
# rotations' condition estimate (Paige-Saunders' Acond) is kept as a net for
# pathological operators only -- it does not track the error closely (2.6 - 3.7 on
# config 4 at 1.4e-7 - 2.9e-7, 4.2 on the weight-0.05 case at 1.3e-6, 16 on a harmless
# weight of 4).
USE_NORMAL_EQUATIONS = True
# Below that guard a float32 solve is not merely slower in the other form -- it is
# outside the contract in BOTH: the float64 oracle (SciPy's algorithm) against float32
# vectors on config 4's blur at 32^3 (tools/_probe/ne_guard_sweep.py,
# profiles/r03_ne_guard_sweep.jsonl), relative weight 0.1 / 0.05 / 0.02 / 0.01, B = gradient:
#     10 iterations   4e-7 / 6e-7..1e-6 / 3.6e-6 / 3.4e-6..4.0e-6
#     20 iterations   3e-7 / 2.8e-5     / 2.2e-4 / 6.3e-4        (both forms alike)
#     32 iterations   3e-7 / 8e-7       / 1.9e-5 / 9e-5
# Once the process has found the dominant eigenvalues its vectors lose orthogonality and,
# until it has converged, an iterate depends on the rounding of every step (in float64
# the same effect is 1e-9, tests/test_gpu_parity.py, ..._at_the_edge_of_its_guard).  So a
# float32 solve with a weak regulariser and more than PROMOTE_FROM_ITERATIONS iterations
# runs its LSMR in float64 (twice the bytes: 0.23 -> ~0.5 s at config 4's size) and hands
# back a float32 result that meets the reference.
PROMOTE_WEAK_REGULARISERS = True
PROMOTE_FROM_ITERATIONS = 10
LAST_PROMOTED = [False]      # (diagnostics)
NE_MIN_WEIGHT = {4: 0.1 * (1 - 1e-9), 8: 1.0e-2}   # by element size
NE_MAX_COND = {4: 1.0e3, 8: 1.0e7}
NE_MAX_ITER = 32
# the blur takes sum |grad y_j|^2 of its input itself (nsol_corr3_wrap_norms_*)
USE_BLUR_NORMS = True
# ... and the whole Lanczos update: two kernels per step (nsol_corr3_wrap_lanczos_a / _b)
# instead of blur, blur and nsol_tk1_lanczos_*; the step's scalars stay on the device
USE_BLUR_LANCZOS = True
# ... also with B = identity (primal-dual deconvolution's prox): with the lean halves
# (ops.LEAN_LANCZOS_HALVES) 82 against 76.6 PD iterations/s at 512^3; with q0 stored it was
# no faster than blur, blur and the element-wise update
LANCZOS_IDENTITY = True
_LAG = 2                     # steps the host's recurrences trail the enqueued kernels
LAST_FORM = [None]           # (diagnostics: "lanczos-in-blur" / "lanczos" / None)
LAST_NE_COND = [None]        # (diagnostics: the estimate of the last run)


def _clipped(x, bounds):
    return x if bounds is None else ops.clip(x, bounds[0], bounds[1], out=x)


def _aliases(t, *others):
    p = t.untyped_storage().data_ptr()
    return any(o is not None and o.untyped_storage().data_ptr() == p
               for o in others)


class MinresCoefficients(object):
    """Paige-Saunders MINRES scalars (scipy.sparse.linalg.minres' recurrences) with
    w_j and x as coefficient vectors over the normalised Lanczos vectors v_j."""

    def __init__(self, capacity, beta1):
        self.oldb, self.beta = 0.0, float(beta1)
        self.dbar, self.epsln, self.phibar = 0.0, 0.0, float(beta1)
        self.cs, self.sn = -1.0, 0.0
        self.w = np.zeros(capacity)
        self.w2 = np.zeros(capacity)
        self.x = np.zeros(capacity)
        self.itn = 0
        self.gmax, self.gmin = 0.0, np.inf
        self.beta1 = float(beta1)
        self.alfas, self.betas = [], []     # T_k: diagonal, off-diagonal (beta_2 ..)

    def step(self, alfa, beta_new):
        """v_{itn} has been multiplied: alfa = v'Mv, beta_new = the next beta."""
        j = self.itn
        self.itn += 1
        self.alfas.append(float(alfa))
        self.betas.append(float(beta_new))
        self.oldb, self.beta = self.beta, float(beta_new)
        oldeps = self.epsln
        delta = self.cs * self.dbar + self.sn * alfa
        gbar = self.sn * self.dbar - self.cs * alfa
        self.epsln = self.sn * self.beta
        self.dbar = -self.cs * self.beta
        gamma = max(math.sqrt(gbar * gbar + self.beta * self.beta),
                    np.finfo(np.float64).eps)
        self.gmax, self.gmin = max(self.gmax, gamma), min(self.gmin, gamma)
        self.cs, self.sn = gbar / gamma, self.beta / gamma
        phi = self.cs * self.phibar
        self.phibar = self.sn * self.phibar
        w1, self.w2 = self.w2, self.w
        e = np.zeros_like(self.w)
        e[j] = 1.0
        self.w = (e - oldeps * w1 - delta * self.w2) / gamma
        self.x = self.x + phi * self.w

    def lsmr_tests(self, normb2, eps=2.220446049250313e-16):
        """The quantities SciPy's LSMR tests after iteration k = itn
        (scipy lsmr.py:416-449), from the Lanczos / MINRES scalars.  With
        x_k = V_k z (orthonormal v_j), Abar = [A; sa B], M = Abar'Abar,
        Abar'b = beta_1 v_1 and V_k' M V_k = T_k = B_k' B_k (the Golub-Kahan
        bidiagonal of the same Krylov space):
            ||Abar' r_k|| = phibar_k                  (LSMR's normar = |zetabar|)
            ||B_k||_F^2   = trace T_k = sum alfa_j    (LSMR's normA^2)
            ||r_k||^2     = ||b||^2 - 2 beta_1 z_1 + z' T_k z     (normr^2)
            ||x_k||       = ||z||
        Returns (test1, test2, t1) = (normr / normb, normar / (normA normr),
        test1 / (1 + normA normx / normb)); eps: the rounding unit of the vectors the
        scalars were summed over."""
        k = self.itn
        z = self.x[:k]
        a = np.asarray(self.alfas[:k])
        quad = float(np.dot(a, z * z))
        if k > 1:
            quad += 2.0 * float(np.dot(np.asarray(self.betas[:k - 1]),
                                       z[:-1] * z[1:]))
        cross = 2.0 * self.beta1 * float(z[0])
        normr2 = normb2 - cross + quad
        # normr^2 is formed by cancellation from sums over vectors of rounding unit
        # `eps` (the Lanczos relation behind it holds to about that): a value within
        # rounding of its terms says nothing -- not even its sign.  SciPy's own
        # estimate stays positive through its recurrences and it would iterate on
        # (atol = btol = 0), so such a residual is reported as unknown (NaN: none of
        # the tests below fires on it) instead of as an exact zero (istop 1).
        if abs(normr2) <= 64.0 * eps * (normb2 + abs(cross) + abs(quad)):
            return float("nan"), float("nan"), float("nan")
        normr = math.sqrt(normr2) if normr2 > 0 else 0.0
        normb = math.sqrt(normb2)
        normA = math.sqrt(float(np.sum(a)))
        normx = float(np.linalg.norm(z))
        test1 = normr / normb
        test2 = self.phibar / (normA * normr) if normA * normr != 0 else np.inf
        t1 = test1 / (1 + normA * normx / normb)
        return test1, test2, t1


def normal_equations_ok(bmode, sa, maxiter, x_like):
    return (USE_NORMAL_EQUATIONS and bmode != ops.B_NONE and sa > 0 and
            1 <= maxiter <= NE_MAX_ITER and
            maxiter + 1 <= _MAX_COMBINED and
            (maxiter + 2) * x_like.numel() * x_like.element_size()
            <= DEFER_X_BYTES)


def lsmr_normal(A, A_adj, b_top, b_bot, bmode, shape, w, sa, x_like, maxiter,
                A_axpby=None, atb=None, x_bounds=None, b_bot_scale=1.0,
                normb2=None, top_norm2=None, g0=None, out_scale=None):
    """The iterates of lsmr_fused from Lanczos on the normal equations (see above).
    top_norm2: a callable that returns |b_top|^2 (like atb: the same in every solve of an
    outer loop around one b).  g0 = (g, |g|^2 as a one-element float64 device tensor): the
    right-hand side vector A^T b_top + sa B^T (b_bot_scale b_bot) when the caller has
    formed it already (ops.admm_vw_update_g: b_bot then holds nothing and is not read).
    b_top is only read, b_bot too.  atb: a callable that returns A^T b_top (a caller
    that solves around the same b again and again keeps it).  x_bounds: see lsmr_fused.
    Returns (x, istop, itn).

    Per step, on the unnormalised Lanczos vector y_j (nothing here needs beta_j):
        t = A y_j with ||t||^2            (the blur's epilogue form)
        sum |grad y_j|^2                   (nsol_tk1_grad_norm_*; B = gradient)
    SciPy's stopping rules (lsmr.py:416-449, atol = btol = 0, conlim = 1e8) after
    every iteration: istop 1 / 2 / 4 / 5 (a residual or a normal-equations residual
    at zero or below machine precision -- a Krylov space exhausted before maxiter)
    are restated on the MINRES scalars (MinresCoefficients.lsmr_tests; normb2 = the
    squared norm of the right-hand side, taken here when the caller does not have
    it); istop 3 / 6 (cond(Abar) estimate >= 1e8 / 1/eps) cannot be reached in this
    form: it hands a run back to the bidiagonalisation -- which carries SciPy's own
    tests -- as soon as its estimate of cond(M) = cond(Abar)^2 exceeds NE_MAX_COND
    (1e3 / 1e7: cond(Abar) <= 32 / 3.2e3).
    -- both sums from the blur itself where A is nsol_amd's one-pass blur on the
    regulariser's grid (nsol_corr3_wrap_norms_*: it holds y_j with a halo anyway) --
        y' = A^T t                         (the blur)
    -- the three scalars of the step (the two sums and beta_j^2 from the previous
    update) travel to the host on a side stream while that last blur runs -- then
        y_{j+1} = (y' + sa^2 K'K y_j)/beta_j - (alfa/beta_j) y_j - (beta_j/beta_{j-1}) y_{j-1}
    with its sum of squares in one pass (nsol_tk1_lanczos_* / nsol_lsmr_v_update_to_*).
    The GPU never waits for the host."""
    import torch
    rho = sa * sa
    n = x_like.numel()
    flat = (n,)
    one = (1.0, 1.0, 1.0)
    grad_mode = bmode == ops.B_GRAD
    # (B = identity or absent: the same kernel with zero weights -- unlike the
    # epilogue form it does not read the tile it overwrites)
    norms, w_norms = None, w if grad_mode else (0.0, 0.0, 0.0)
    if USE_BLUR_NORMS and (not grad_mode or (
            len(shape) == 3 and
            tuple(getattr(A_axpby, "shape", ())) == tuple(shape))):
        norms = getattr(A_axpby, "norms", None)
    # g = A^T b_top + sa B^T b_bot
    if g0 is None:
        atu = atb() if atb is not None else A_adj(b_top)
        g = torch.empty_like(x_like)
    else:
        atu, g = None, g0[0]

    def rhs_norm2():
        # (right-hand side [b_top; b_bot_scale * b_bot]; the caller's figure if it has one)
        if normb2 is not None:            # (a callable: the caller's figure, fetched late)
            return normb2() if callable(normb2) else normb2
        r = top_norm2() if top_norm2 is not None else ops.dot(b_top, b_top)
        if b_bot is not None and bmode != ops.B_NONE:
            r += b_bot_scale ** 2 * ops.dot(b_bot, b_bot)
        return r
    halves = getattr(A_axpby, "lanczos", None) if USE_BLUR_LANCZOS else None
    # (B = identity: the kernels take it -- rho_ident; with q0 stored the three-kernel
    # form was as fast, with the lean halves primal-dual deconvolution at 512^3 runs
    # 0.122 s per ten iterations against 0.131 s; LANCZOS_IDENTITY = False for the A/B)
    if halves is not None and (grad_mode or (LANCZOS_IDENTITY and
                                              bmode == ops.B_IDENTITY)) and \
            (not grad_mode or (tuple(w) == (1.0, 1.0, 1.0) and
                               tuple(getattr(A_axpby, "shape", ())) == tuple(shape))):
        # both halves of every step inside the blur: |g|^2 goes straight onto the
        # device's scalar board and the steps are enqueued behind it, unseen by the host
        lb = ops.LanczosBoard(x_like, maxiter, rho if grad_mode else 0.0,
                              0.0 if grad_mode else rho)
        if g0 is None:
            ops.lsmr_v_update(atu, b_bot, atu, bmode, shape, w, 1.0, sa * b_bot_scale, 0.0,
                              out=g, result=lb.slot_norm2(0))
        else:
            lb.slot_norm2(0).copy_(g0[1])
        del atu
        got = _lanczos_in_blur(halves, lb, g, rho, x_like, maxiter, rhs_norm2, x_bounds,
                               out_scale=out_scale)
        if got is not None:
            return got
        beta1 = math.sqrt(float(lb.board[0].item()))     # (the kernels do not apply)
    else:
        if g0 is None:
            beta1 = math.sqrt(ops.lsmr_v_update(atu, b_bot, atu, bmode, shape, w, 1.0,
                                                sa * b_bot_scale, 0.0, out=g))
        else:
            beta1 = math.sqrt(float(g0[1].item()))
        del atu
    if beta1 == 0:
        return _clipped(torch.zeros_like(x_like), x_bounds), 0, 0
    normb2 = rhs_norm2()

    def scipy_stop(co):
        return _scipy_stop(co, normb2, float(torch.finfo(x_like.dtype).eps))
    LAST_FORM[0] = "lanczos"
    ys, betas = [g], [beta1]
    co = MinresCoefficients(maxiter + 1, beta1)
    t = torch.zeros_like(x_like)
    slots = torch.zeros(3, dtype=torch.float64, device=x_like.device)
    fetch = ops.scalar_fetchers(x_like.device, 3, 1)[0]
    istop = 7
    alfa_prev = None
    for itn in range(1, maxiter + 1):
        yj = ys[-1]
        got, have_gg = None, False
        if norms is not None:
            got = norms(yj, t, w_norms, slots[0:2])
            have_gg = got is not None
        if got is None and A_axpby is not None:
            got = A_axpby(yj, t, 1.0, 0.0, result=slots[0:1])
        if got is None:
            t = A(yj)
            ops.lsmr_v_update(t, None, t, ops.B_NONE, flat, one, 1.0, 0.0, 0.0,
                              out=torch.empty_like(t), result=slots[0:1])
        if grad_mode and not have_gg:
            ops.tk1_grad_norm(yj, shape, w, result=slots[1:2])
        fetch.start(slots)
        yp = A_adj(t)                                    # A^T A y_j
        vals = fetch.wait()
        tt, gg, nb2 = float(vals[0]), float(vals[1]), float(vals[2])
        if alfa_prev is not None:
            beta_j = math.sqrt(nb2) if nb2 > 0 else 0.0
            co.step(alfa_prev, beta_j)
            stop = scipy_stop(co)
            if stop == 0 and (beta_j == 0 or not math.isfinite(beta_j)):
                stop = 2                                  # Krylov space exhausted
            if stop:
                ys.pop()              # (the vector this step was working on)
                alfa_prev = None
                istop = stop
                break
            betas.append(beta_j)
        beta = betas[-1]
        if itn == 1 and rho < NE_MIN_WEIGHT[x_like.element_size()] * tt / (beta * beta):
            return None, -1, 0         # regulariser too weak against ||A||^2
        prev = ys[-2] if len(ys) >= 2 else None
        c_prev = -beta / betas[-2] if prev is not None else 0.0
        ynew = torch.empty_like(x_like)
        if grad_mode:
            alfa = (tt + rho * gg) / (beta * beta)
            ops.tk1_lanczos(yj, yp, prev, shape, w, rho / beta, 1.0 / beta,
                            -alfa / beta, c_prev, out=ynew, result=slots[2:3])
        else:                                            # B = identity: B'B y_j = y_j
            alfa = tt / (beta * beta) + rho
            ops.lsmr_v_update(yj, prev, yp,
                              ops.B_IDENTITY if prev is not None else ops.B_NONE,
                              flat, one, (rho - alfa) / beta, c_prev, 1.0 / beta,
                              out=ynew, result=slots[2:3])
        del yp
        ys.append(ynew)
        alfa_prev = alfa
    if alfa_prev is not None:                            # the last step's beta
        nb2 = float(slots[2].item())
        co.step(alfa_prev, math.sqrt(nb2) if nb2 > 0 else 0.0)
        istop = scipy_stop(co) or istop      # (SciPy: 7, overridden by 5 / 4 / 2 / 1)
    k = co.itn
    LAST_NE_COND[0] = co.gmax / co.gmin if co.gmin > 0 else np.inf
    if LAST_NE_COND[0] > NE_MAX_COND[x_like.element_size()]:
        return None, -1, k             # too ill-conditioned for this form
    x = ops.lincomb_many(ys[:k], [co.x[j] / betas[j] for j in range(k)],
                         bounds=x_bounds)
    return x, istop, k


def _scipy_stop(co, normb2, eps=2.220446049250313e-16):
    """istop of scipy lsmr.py:432-449 after iteration co.itn (0: go on), atol = btol =
    0: the tests on machine precision (4, 5) and on exact zeros (1, 2).  eps: see
    MinresCoefficients.lsmr_tests."""
    test1, test2, t1 = co.lsmr_tests(normb2, eps)
    stop = 0
    if 1 + test2 <= 1:
        stop = 5
    if 1 + t1 <= 1:
        stop = 4
    if test2 <= 0.0:
        stop = 2
    if test1 <= 0.0:
        stop = 1
    return stop


def _lanczos_in_blur(halves, lb, g, rho, x_like, maxiter, rhs_norm2, x_bounds,
                     out_scale=None):
    """lsmr_normal's loop with both halves of every step inside the blur
    (nsol_corr3_wrap_lanczos_a / _b, nsol_blur3_dma.hpp): per step
        t = A y_j, |t|^2, |grad y_j|^2, q0 = (rho / beta_j) K'K y_j - (beta_j / beta_{j-1}) y_{j-1}
        y_{j+1} = (1 / beta_j) A t + q0 - (alfa_j / beta_j) y_j, |y_{j+1}|^2
    with the coefficients formed ON THE DEVICE from the sums (lb: ops.LanczosBoard,
    |g|^2 already on its way to board[0]): the steps are enqueued back to back.  The
    host follows _LAG steps behind: the four sums of step j travel to pinned memory on
    a side stream once step j is enqueued, and are turned into MINRES' recurrences and
    SciPy's stopping tests while the GPU runs steps j + 1 .. j + _LAG -- at the end only
    the last steps' arithmetic (a few tens of microseconds) stands between the last
    kernel and the assembly of x.  A stop seen that way ends the enqueuing (the vectors
    already on their way are not used); the guard on the regulariser's weight is step
    0's first check.  Returns (x, istop, itn), (None, -1, 0) when the guard fails, None
    when the kernels do not apply (nothing but the board's initialisation was
    enqueued)."""
    import torch
    half_a, half_b = halves
    lb.init()
    ys = [g]
    # (the lean halves keep no q0 array: the second half forms the step's K'K y itself)
    t = torch.empty_like(x_like)
    q0 = None if getattr(half_a, "lean", False) else torch.empty_like(x_like)
    assert maxiter >= 1
    fetchers = ops.scalar_fetchers(x_like.device, 4, _LAG + 1)
    state = {"co": None, "betas": None, "istop": 7, "normb2": None, "verdict": None}

    def digest(j, v):
        """Step j's sums (|y_j|^2, |A y_j|^2, |grad y_j|^2, |y_{j+1}|^2): True = stop."""
        nb2, tt, gg, nb2n = (float(q) for q in v[:4])
        if j == 0:
            if not nb2 > 0:                                 # g = 0: x = 0
                state["verdict"] = "zero"
                return True
            if rho < NE_MIN_WEIGHT[x_like.element_size()] * tt / nb2:
                state["verdict"] = "weak"                    # regulariser too weak
                return True
            state["betas"] = [math.sqrt(nb2)]
            state["co"] = MinresCoefficients(maxiter + 1, state["betas"][0])
            state["normb2"] = rhs_norm2()
        co = state["co"]
        alfa = (tt + lb.rho_grad * gg) / nb2 + lb.rho_ident
        beta_next = math.sqrt(nb2n) if nb2n > 0 else 0.0
        co.step(alfa, beta_next)
        stop = _scipy_stop(co, state["normb2"], float(torch.finfo(x_like.dtype).eps))
        if j + 1 < maxiter:
            if stop == 0 and (beta_next == 0 or not math.isfinite(beta_next)):
                stop = 2                                  # Krylov space exhausted
            if stop:
                state["istop"] = stop
                return True
            state["betas"].append(beta_next)
        else:
            state["istop"] = stop or state["istop"]
        return False

    done, stopped = 0, False
    lean = getattr(half_a, "lean", False)
    for j in range(maxiter):
        # (the last step's vector is never read -- x is assembled from y_0 .. y_{k-1} --
        # only its norm, the next beta, is: the lean second half then stores nothing)
        ynew = None if (lean and j == maxiter - 1) else torch.empty_like(x_like)
        if not half_a(ys[-1], ys[-2] if j > 0 else None, t, q0, lb, j):
            if j == 0:
                return None
            raise RuntimeError("nsol_corr3_wrap_lanczos_a stopped applying mid-solve")
        if not half_b(t, q0, ys[-1], ynew, lb, j):
            raise RuntimeError("nsol_corr3_wrap_lanczos_b does not apply")
        ys.append(ynew)
        fetchers[j % (_LAG + 1)].start(lb.board[3 * j:3 * j + 4])
        if j == 0 and rho < NE_MIN_WEIGHT[x_like.element_size()]:
            # (a weight that the guard may refuse -- it holds |A y_0|^2 / |y_0|^2 against
            # it, at most |A|^2: step 0's sums are waited for before more is enqueued; a
            # refused solve used to cost the _LAG steps already on their way)
            stopped = digest(0, fetchers[0].wait())
            done = 1
            if stopped:
                break
        elif j >= _LAG:
            stopped = digest(done, fetchers[done % (_LAG + 1)].wait())
            done += 1
            if stopped:
                break
    last = len(ys) - 1                 # steps enqueued
    while not stopped and done < last:
        stopped = digest(done, fetchers[done % (_LAG + 1)].wait())
        done += 1
    # (an early stop leaves copies of later steps' sums in flight on the side stream:
    # they read lb.board, which is freed with this frame -- let them finish first)
    for j in range(done, last):
        fetchers[j % (_LAG + 1)].wait()
    if state["verdict"] == "zero":
        return _clipped(torch.zeros_like(x_like), x_bounds), 0, 0
    if state["verdict"] == "weak":
        return None, -1, 0
    co, betas = state["co"], state["betas"]
    k = co.itn
    LAST_NE_COND[0] = co.gmax / co.gmin if co.gmin > 0 else np.inf
    LAST_FORM[0] = "lanczos-in-blur"
    if LAST_NE_COND[0] > NE_MAX_COND[x_like.element_size()]:
        return None, -1, k
    s = 1.0
    if out_scale is not None and out_scale[0] > 0 and math.isfinite(out_scale[0]):
        s = float(out_scale[0])
        out_scale[1] = True
        if x_bounds is not None:
            x_bounds = (x_bounds[0] * s, x_bounds[1] * s)
    x = ops.lincomb_many(ys[:k], [s * co.x[j] / betas[j] for j in range(k)],
                         bounds=x_bounds)
    return x, state["istop"], k


def _operators_in_float64(A, A_adj, x_like):
    """(A, A_adj) for a float32 solve that is promoted to float64, or None when the
    caller's operators cannot follow: a foreign NumPy callable behind a float32
    BridgedCallable is bridged in float64 instead; a device operator is asked once
    (a zero vector through A and A_adj) whether it answers float64 with float64 --
    one bound to float32 tensors (dense taps uploaded in float32, a caller's own
    float32 kernel) does not, and the solve then stays in float32."""
    import torch
    from .bridge import BridgedCallable, _is_gpu_failure

    def widen(f):
        return BridgedCallable(f.fn, np.float64) if isinstance(f, BridgedCallable) else f
    A64, At64 = widen(A), widen(A_adj)
    n = x_like.numel()
    try:
        r = A64(torch.zeros(n, dtype=torch.float64, device=x_like.device))
        if not (isinstance(r, torch.Tensor) and r.is_cuda and r.dtype == torch.float64):
            return None
        r = At64(r.contiguous().view(-1))
        if not (isinstance(r, torch.Tensor) and r.is_cuda and
                r.dtype == torch.float64 and r.numel() == n):
            return None
    except (TypeError, AttributeError, ValueError, RuntimeError) as e:
        if _is_gpu_failure(e):
            raise
        return None
    return A64, At64


def lsmr_fused(A, A_adj, b_top, b_bot, bmode, shape, w, sa, x_like, maxiter,
               atol=0.0, btol=0.0, conlim=1e8, A_axpby=None, normb2=None, top_norm2=None,
               own_b=True, atb=None, x_bounds=None, b_bot_scale=1.0,
               allow_normal=True, g0=None, b_bot_fill=None, out_scale=None):
    """Same algorithm for the augmented system [A; sa*B] with B in {none, grad,
    identity}, on the fused kernels of nsol_lsmr.hip.  The Golub-Kahan vectors
    are held unnormalised (ut = su*u, vt = sv*v); b_top / b_bot are consumed.
    A, A_adj: device callables (flat tensor -> flat tensor).  A_axpby(v, io, ca,
    cb) -> sum of squares or None: io = ca * A v + cb * io formed by the blur
    itself (its epilogue), when A is nsol_amd's one-pass blur.  normb2: the squared
    norm of the right-hand side when the caller has it already (or a callable that
    returns it: asked for when the stopping tests first need it, not before).  own_b: b_top may be
    consumed (False: it is the caller's and gets copied where the bidiagonalisation
    overwrites it).  atb: see lsmr_normal.  x_bounds = (lo, hi): the solution comes back
    projected onto them (tikhonov_linear_solver.py:142-143 applied to the result), in
    the pass that assembles it from the stored vectors where there is one.
    b_bot_scale: the lower block of the right-hand side is b_bot_scale * b_bot (the
    caller's array is then only read: the normal-equations form folds the factor into a
    coefficient, the bidiagonalisation scales into a copy).  g0: see lsmr_normal; b_bot
    then holds nothing until b_bot_fill() has written it -- called here when the solve
    cannot stay with the normal equations and needs the block itself.  out_scale =
    [s, False], s > 0: the caller wants s * x; where x is assembled from stored vectors the
    factor goes into their coefficients (and the bounds) and out_scale[1] becomes True --
    otherwise x comes back as it is and the caller multiplies."""
    import torch
    wide = None
    if bmode == ops.B_NONE and PROMOTE_WEAK_REGULARISERS and x_like.element_size() == 4 \
            and maxiter > PROMOTE_FROM_ITERATIONS:
        # (no regulariser at all: the same, more so)
        wide = _operators_in_float64(A, A_adj, x_like)
    if wide is not None:
        LAST_PROMOTED[0] = True
        x64, istop, itn = lsmr_fused(
            wide[0], wide[1], b_top.double(), None, bmode, shape, w, sa, x_like.double(),
            maxiter, atol=atol, btol=btol, conlim=conlim, A_axpby=A_axpby, own_b=True,
            x_bounds=x_bounds)
        return x64.to(x_like.dtype), istop, itn
    if atol == 0.0 and btol == 0.0 and allow_normal and \
            normal_equations_ok(bmode, sa, maxiter, x_like):
        x, istop, itn = lsmr_normal(A, A_adj, b_top, b_bot, bmode, shape, w, sa,
                                    x_like, maxiter, A_axpby=A_axpby, atb=atb,
                                    x_bounds=x_bounds, b_bot_scale=b_bot_scale,
                                    normb2=normb2, top_norm2=top_norm2, g0=g0,
                                    out_scale=out_scale)
        if x is not None:
            return x, istop, itn
        # (the weight is below the guard or the condition estimate came out too high:
        # nothing was consumed)
        if g0 is not None and b_bot_fill is not None:
            b_bot_fill()                  # (the lower block itself is needed from here on)
            b_bot_fill = None
        if PROMOTE_WEAK_REGULARISERS and x_like.element_size() == 4 and \
                maxiter > PROMOTE_FROM_ITERATIONS:
            wide = _operators_in_float64(A, A_adj, x_like)
        if wide is not None:
            # float32 vectors cannot hold the 1e-5 contract there, in EITHER form
            # (see PROMOTE_WEAK_REGULARISERS): this solve runs in float64, and as the
            # bidiagonalisation -- SciPy's own recurrence: in this regime an iterate
            # depends on the form at the 1e-6 level even in float64
            LAST_PROMOTED[0] = True
            x64, istop, itn = lsmr_fused(
                wide[0], wide[1], b_top.double(),
                None if b_bot is None else b_bot.double(),
                bmode, shape, w, sa, x_like.double(), maxiter, atol=atol, btol=btol,
                conlim=conlim, A_axpby=A_axpby, normb2=None, own_b=True, atb=None,
                x_bounds=x_bounds, b_bot_scale=b_bot_scale, allow_normal=False)
            return x64.to(x_like.dtype), istop, itn
    if g0 is not None and b_bot_fill is not None:
        b_bot_fill()
    if not own_b:                 # (the caller's b: consumed below, so work in a copy)
        b_top = b_top.clone()
    if b_bot is not None and b_bot_scale != 1.0:
        b_bot = ops.scale(b_bot, b_bot_scale)
    ut, ub = b_top, b_bot
    if normb2 is not None:           # ||[b_top; b_bot]||^2 known to the caller
        normb = math.sqrt(normb2() if callable(normb2) else normb2)
    else:
        normb = math.sqrt(ops.dot(ut, ut) +
                          (ops.dot(ub, ub) if ub is not None else 0.0))
    defer = (DEFER_X and maxiter + 1 <= _MAX_COMBINED and
             (maxiter + 1) * x_like.numel() * x_like.element_size()
             <= DEFER_X_BYTES)
    x = None if defer else torch.zeros_like(x_like)
    beta = normb
    su = beta if beta > 0 else 1.0
    vts, svs = [], []
    slots = torch.zeros(2, dtype=torch.float64, device=x_like.device)

    def next_v(atu, v_old, c_atu, c_btu, c_v):
        """(new vt, its squared norm).  When every v_k is kept the new vector
        goes to a buffer of its own (the same bytes move as for the update in
        place) -- never into the one A^T u came in: a caller's operator may hand
        back its argument or a buffer it reuses from call to call."""
        if not defer:
            return v_old, ops.lsmr_v_update(atu, ub, v_old, bmode, shape, w,
                                            c_atu, c_btu, c_v)
        out = torch.empty_like(x_like)
        return out, ops.lsmr_v_update(atu, ub, v_old, bmode, shape, w, c_atu,
                                      c_btu, c_v, out=out)

    if beta > 0:
        # vt = A^T ut + sa B^T ub  (raw, = su * A^T u)
        atu = A_adj(ut)
        if defer:
            vt, nv2 = next_v(atu, atu, 1.0, sa, 0.0)
        else:
            vt, nv2 = next_v(atu, torch.zeros_like(x_like), 1.0, sa, 0.0)
        alpha = math.sqrt(nv2) / su
    else:
        vt = torch.zeros_like(x_like)
        alpha = 0.0
    sv = su * alpha if alpha > 0 else 1.0
    if defer:
        vts.append(vt)
        svs.append(sv)
        coef = SolutionCoefficients(maxiter + 1)

    itn = 0
    zetabar = alpha * beta
    alphabar = alpha
    rho = rhobar = cbar = 1.0
    sbar = 0.0
    if not defer:
        h = ops.scale(vt, 1.0 / sv)
        hbar = torch.zeros_like(x_like)
    betadd, betad = beta, 0.0
    rhodold = 1.0
    tautildeold = thetatilde = zeta = d = 0.0
    normA2 = alpha * alpha
    maxrbar, minrbar = 0.0, 1e100
    istop = 0
    ctol = 1.0 / conlim if conlim > 0 else 0.0
    if alpha * beta == 0 or normb == 0:
        return _clipped(torch.zeros_like(x_like) if defer else x, x_bounds), istop, itn

    while itn < maxiter:
        itn += 1
        # ut <- A v - alpha u   (v = vt/sv, u = ut/su).  The two sums of squares
        # stay on the device until A^T ut -- which needs no scalar -- is enqueued:
        # one read-back for both, hidden behind that blur
        top = None
        if A_axpby is not None and bmode != ops.B_NONE:
            top = A_axpby(vt, ut, 1.0 / sv, -alpha / su, result=slots[0:1])
        if top is not None:
            ops.lsmr_u_update(None, vt, ut, ub, bmode, shape, w, 1.0 / sv,
                              sa / sv, -alpha / su, result=slots[1:2])
        else:
            ops.lsmr_u_update(A(vt), vt, ut, ub, bmode, shape, w, 1.0 / sv,
                              sa / sv, -alpha / su, result=slots[1:2])
        atu = A_adj(ut)
        sums = slots.cpu()
        nu2 = float(sums[1]) + (float(sums[0]) if top is not None else 0.0)
        beta = math.sqrt(nu2)
        su = beta if beta > 0 else 1.0
        if beta > 0:
            # vt <- A^T u - beta v
            vt, nv2 = next_v(atu, vt, 1.0 / beta, sa / beta, -beta / sv)
            alpha = math.sqrt(nv2)
            sv = alpha if alpha > 0 else 1.0
            if defer:
                vts.append(vt)
                svs.append(sv)

        chat, shat, alphahat = _sym_ortho(alphabar, 0.0)
        rhoold = rho
        c, s, rho = _sym_ortho(alphahat, beta)
        thetanew = s * alpha
        alphabar = c * alpha
        rhobarold, zetaold = rhobar, zeta
        thetabar = sbar * rho
        rhotemp = cbar * rho
        cbar, sbar, rhobar = _sym_ortho(cbar * rho, thetanew)
        zeta = cbar * zetabar
        zetabar = -sbar * zetabar

        c_hbar = -(thetabar * rho / (rhoold * rhobarold))
        c_x = zeta / (rho * rhobar)
        c_h = -(thetanew / rho)
        if defer:
            # hbar = h + c_hbar hbar; x += c_x hbar; h = v_new + c_h h -- on the
            # coefficients; ||x||^2 = sum a_k^2 for orthonormal v_k (it only enters
            # the stopping test that fires when the residual is exactly zero)
            coef.step(c_hbar, c_x, c_h, len(vts) - 1)
            normx2 = coef.normx2()
        else:
            normx2 = ops.lsmr_hx_update(hbar, x, h, vt, c_hbar, c_x, c_h,
                                        1.0 / sv)

        betaacute = chat * betadd
        betacheck = -shat * betadd
        betahat = c * betaacute
        betadd = -s * betaacute
        thetatildeold = thetatilde
        ctildeold, stildeold, rhotildeold = _sym_ortho(rhodold, thetabar)
        thetatilde = stildeold * rhobar
        rhodold = ctildeold * rhobar
        betad = -stildeold * betad + ctildeold * betahat
        tautildeold = (zetaold - thetatildeold * tautildeold) / rhotildeold
        taud = (zeta - thetatilde * tautildeold) / rhodold
        d = d + betacheck * betacheck
        normr = math.sqrt(d + (betad - taud) ** 2 + betadd * betadd)
        normA2 = normA2 + beta * beta
        normA = math.sqrt(normA2)
        normA2 = normA2 + alpha * alpha
        maxrbar = max(maxrbar, rhobarold)
        if itn > 1:
            minrbar = min(minrbar, rhobarold)
        condA = max(maxrbar, rhotemp) / min(minrbar, rhotemp)

        normar = abs(zetabar)
        normx = math.sqrt(normx2)
        test1 = normr / normb
        test2 = normar / (normA * normr) if (normA * normr) != 0 else np.inf
        test3 = 1.0 / condA
        t1 = test1 / (1 + normA * normx / normb)
        rtol = btol + atol * normA * normx / normb
        if itn >= maxiter:
            istop = 7
        if 1 + test3 <= 1:
            istop = 6
        if 1 + test2 <= 1:
            istop = 5
        if 1 + t1 <= 1:
            istop = 4
        if test3 <= ctol:
            istop = 3
        if test2 <= atol:
            istop = 2
        if test1 <= rtol:
            istop = 1
        if istop > 0:
            break
    if defer:
        x = ops.lincomb_many(vts, [coef.x[j] / svs[j] for j in range(len(vts))],
                             bounds=x_bounds)
    else:
        x = _clipped(x, x_bounds)
    return x, istop, itn
