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


def lsmr_fused(A, A_adj, b_top, b_bot, bmode, shape, w, sa, x_like, maxiter,
               atol=0.0, btol=0.0, conlim=1e8, A_axpby=None, normb2=None):
    """Same algorithm for the augmented system [A; sa*B] with B in {none, grad,
    identity}, on the fused kernels of nsol_lsmr.hip.  The Golub-Kahan vectors
    are held unnormalised (ut = su*u, vt = sv*v); b_top / b_bot are consumed.
    A, A_adj: device callables (flat tensor -> flat tensor).  A_axpby(v, io, ca,
    cb) -> sum of squares or None: io = ca * A v + cb * io formed by the blur
    itself (its epilogue), when A is nsol_amd's one-pass blur.  normb2: the squared
    norm of the right-hand side when the caller has it already."""
    import torch
    ut, ub = b_top, b_bot
    if normb2 is not None:           # ||[b_top; b_bot]||^2 known to the caller
        normb = math.sqrt(normb2)
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
        return (torch.zeros_like(x_like) if defer else x), istop, itn

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
        x = ops.lincomb_many(vts, [coef.x[j] / svs[j] for j in range(len(vts))])
    return x, istop, itn
