"""CPU oracle for the NSoL primal-dual / ADMM hot path.

TEST INFRASTRUCTURE ONLY.  This module is a NumPy float64 restatement of the
reference algorithm (gift-surg/NSoL v0.1.14).  It is imported only by
``tests/``, by ``__graft_entry__.smoke()`` and by the ``cpu_baseline`` leg of
``bench.py`` -- never by the product package ``nsol_amd`` (which must fail
loudly if its HIP library is missing).

Parity pin: every function below is checked in ``tests/test_oracle_golden.py``
against golden vectors in ``tests/golden/*.npz`` that were produced by importing
the reference itself in the build container (``tools/make_goldens.py``).

The restatement uses explicit slicing instead of ``scipy.ndimage.convolve``
(section A of SURVEY.md).  The ``*_refstyle`` functions at the bottom repeat
the per-iteration operation sequence in the reference's own style (ndimage
correlate calls + concatenate + flatten copies); they exist so that the CPU
baseline that is timed next to the GPU does the same amount of work as the
reference does.

Array conventions (reference ``linear_operators.py:121-144``): N-D arrays are
indexed ``[z, y, x]``; ``spacing[0]`` scales the LAST array axis (x); the
gradient field is stacked on axis 0: ``(dim*Z, Y, X) = [d/dx; d/dy; d/dz]``.
"""

import numpy as np

EPS = 1e-10  # reference definitions.py:11


# --------------------------------------------------------------------------
# Stencil taps (reference kernels.py)
# --------------------------------------------------------------------------
def _check_spacing(dimension, spacing):
    # kernels.py:18-26
    spacing = np.atleast_1d(spacing).astype(float)
    if spacing.size != dimension:
        raise ValueError("dimension of spacing and space must be the same")
    return spacing


def gaussian_taps(dimension, cov, spacing=None, alpha_cut=3):
    """Dense normalised Gaussian tap array exactly as the reference builds it.

    1D: kernels.py:80-100; 2D: kernels.py:120-158; 3D: kernels.py:198-238.
    Reproduces the reference's axis quirk for anisotropic covariances: the tap
    array is reshaped (x_size, y_size[, z_size]) and later applied to an array
    indexed [(z,) y, x].
    """
    spacing = _check_spacing(
        dimension, np.ones(dimension) if spacing is None else spacing)
    if dimension == 1:
        cov = float(np.asarray(cov).reshape(-1)[0])
        x_max = np.ceil(np.sqrt(cov) * alpha_cut / spacing)
        pts = np.arange(-x_max[0], x_max[0] + 1, 1)
        w = np.exp(-0.5 * pts * (spacing[0] ** 2 / cov) * pts)
        return w / np.sum(w)

    cov = np.asarray(cov, dtype=float)
    if cov.shape != (dimension, dimension):
        raise ValueError("Numpy array 'cov' must be of shape (%d,%d)" %
                         (dimension, dimension))
    ext = np.ceil(np.sqrt(cov.diagonal()) * alpha_cut / spacing)
    intervals = [np.arange(-e, e + 1, 1) for e in ext]  # x, y(, z)
    grids = np.meshgrid(*intervals, indexing='ij')      # X, Y(, Z)
    # points are stacked (Z,) Y, X  -- kernels.py:143, 223
    pts = np.array([g.flatten() for g in grids[::-1]])
    S = np.diag(spacing)
    cinv = S.dot(np.linalg.inv(cov)).dot(S)
    vals = np.sum(pts * cinv.dot(pts), 0)
    w = np.exp(-0.5 * vals)
    w = w / np.sum(w)
    return w.reshape(*[iv.size for iv in intervals])


# --------------------------------------------------------------------------
# Finite differences (reference linear_operators.py:98-169, kernels.py:102-286)
# --------------------------------------------------------------------------
def _axis_of(dimension, a):
    """Array axis acted on by direction a (0=x,1=y,2=z): the LAST axis is x."""
    return dimension - 1 - a


def d_forward(x, axis, h):
    """(D u)[i] = (u[i+1] - u[i]) / h, u := 0 past the last index.

    = ndimage.convolve(u, [1,-1]/h, mode='constant') (linear_operators.py:103).
    """
    w = 1.0 / h            # the reference scales the taps: [1, -1] / h
    out = x * (-w)
    sl_dst = [slice(None)] * x.ndim
    sl_src = [slice(None)] * x.ndim
    sl_dst[axis] = slice(0, -1)
    sl_src[axis] = slice(1, None)
    out[tuple(sl_dst)] += x[tuple(sl_src)] * w
    return out


def d_forward_adj(p, axis, h):
    """(D^T p)[i] = (p[i-1] - p[i]) / h, p[-1] := 0.

    = ndimage.convolve(p, -[0,1,-1]/h, mode='constant') (linear_operators.py:104).
    """
    w = 1.0 / h
    out = p * (-w)
    sl_dst = [slice(None)] * p.ndim
    sl_src = [slice(None)] * p.ndim
    sl_dst[axis] = slice(1, None)
    sl_src[axis] = slice(0, -1)
    out[tuple(sl_dst)] += p[tuple(sl_src)] * w
    return out


def grad(x, spacing=None):
    """K = nabla, stacked on axis 0 (linear_operators.py:121-144)."""
    d = x.ndim
    spacing = _check_spacing(d, np.ones(d) if spacing is None else spacing)
    parts = [d_forward(x, _axis_of(d, a), spacing[a]) for a in range(d)]
    if d == 1:
        return parts[0]
    return np.concatenate(parts)


def grad_adj(p, spacing=None):
    """K^T, exact transpose of grad (linear_operators.py:158-169)."""
    d = p.ndim
    spacing = _check_spacing(d, np.ones(d) if spacing is None else spacing)
    if d == 1:
        return d_forward_adj(p, 0, spacing[0])
    parts = np.array_split(p, d)
    out = d_forward_adj(parts[0], _axis_of(d, 0), spacing[0])
    for a in range(1, d):
        out += d_forward_adj(parts[a], _axis_of(d, a), spacing[a])
    return out


# --------------------------------------------------------------------------
# Dense N-D convolution with ndimage boundary modes
# (reference linear_operators.py:60-68 -> scipy.ndimage.convolve)
# --------------------------------------------------------------------------
_PAD_MODE = {"wrap": "wrap", "constant": "constant", "nearest": "edge",
             "reflect": "symmetric", "mirror": "reflect"}


def _pad_axis(x, axis, lo, hi, mode):
    """np.pad along one axis, repeating the pad when it exceeds the length
    (np.pad handles over-long wrap/symmetric pads by iterating)."""
    pw = [(0, 0)] * x.ndim
    pw[axis] = (lo, hi)
    if mode == "constant":
        return np.pad(x, pw, mode="constant", constant_values=0)
    return np.pad(x, pw, mode=_PAD_MODE[mode])


def convolve_nd(x, kernel, mode="wrap"):
    """out[i] = sum_j kernel[j] * x[i + c' - j] with scipy.ndimage.convolve's
    centre convention: correlate with the reversed kernel, the centre of the
    reversed kernel being size//2 for odd and size//2 - 1 for even sizes."""
    x = np.asarray(x, dtype=np.float64)
    kernel = np.asarray(kernel, dtype=np.float64)
    if kernel.ndim != x.ndim:
        raise RuntimeError("filter weights array has incorrect shape.")
    w = kernel[tuple([slice(None, None, -1)] * kernel.ndim)]
    xp = x
    centres = []
    for ax in range(x.ndim):
        s = w.shape[ax]
        c = s // 2 - (1 if s % 2 == 0 else 0)
        centres.append(c)
        xp = _pad_axis(xp, ax, c, s - 1 - c, mode)
    out = np.zeros_like(x)
    for idx in np.ndindex(*w.shape):
        wt = w[idx]
        if wt == 0.0:
            continue
        sl = tuple(slice(idx[ax], idx[ax] + x.shape[ax])
                   for ax in range(x.ndim))
        out += wt * xp[sl]
    return out


def separable_factors(kernel, tol=1e-13):
    """If the dense tap array is an outer product of per-axis vectors return
    them (one per array axis), else None.  For a normalised rank-1 array the
    marginals are exactly the factors."""
    kernel = np.asarray(kernel, dtype=np.float64)
    if kernel.ndim == 1:
        return [kernel]
    total = kernel.sum()
    facs = []
    for ax in range(kernel.ndim):
        other = tuple(a for a in range(kernel.ndim) if a != ax)
        facs.append(kernel.sum(axis=other))
    outer = facs[0]
    for f in facs[1:]:
        outer = np.multiply.outer(outer, f)
    outer = outer / total ** (kernel.ndim - 1)
    if np.max(np.abs(outer - kernel)) <= tol * np.max(np.abs(kernel)):
        facs[0] = facs[0] / total ** (kernel.ndim - 1)
        return facs
    return None


def gaussian_blur(x, cov, spacing=None, alpha_cut=3):
    """A = A^T (linear_operators.py:82-86): periodic dense convolution."""
    taps = gaussian_taps(x.ndim, cov, spacing, alpha_cut)
    return convolve_nd(x, taps, mode="wrap")


# --------------------------------------------------------------------------
# Proximal operators (reference proximal_operators.py)
# --------------------------------------------------------------------------
def prox_tv_conj(x, sigma):
    # proximal_operators.py:138-140 (element-wise, sigma unused)
    return x / np.maximum(1, np.abs(x))


def prox_huber_conj(x, sigma, gamma=0.05):
    # proximal_operators.py:156-159 (the reference divides in place)
    x = x / (1. + sigma * gamma)
    return x / np.maximum(1, np.abs(x))


def prox_ell1_denoising(x, tau, x0, x_scale=1.):
    # proximal_operators.py:95-98
    x0 = x0 / float(x_scale)
    return x0 + np.maximum(np.abs(x - x0) - tau, 0) * np.sign(x - x0)


def prox_ell2_denoising(x, tau, x0, x_scale=1.):
    # proximal_operators.py:117-120
    x0 = x0 / float(x_scale)
    return (x + tau * x0) / (1. + tau)


# --------------------------------------------------------------------------
# Loss functions (reference loss_functions.py:82-266)
# --------------------------------------------------------------------------
def loss(name, f2, f_scale=1.):
    s2 = float(f_scale * f_scale)
    z = f2 / s2
    if name == "linear":
        return f2
    if name == "soft_l1":
        return 2. * (np.sqrt(1. + z) - 1.) * s2
    if name == "huber":
        g = 1.345
        return np.where(z < g * g, z, 2. * g * np.sqrt(z) - g * g) * s2
    if name == "cauchy":
        return np.log1p(z) * s2
    if name == "arctan":
        return np.arctan(z) * s2
    raise ValueError(name)


def gradient_loss(name, f2, f_scale=1.):
    s2 = float(f_scale * f_scale)
    z = f2 / s2
    if name == "linear":
        return np.ones_like(f2).astype(np.float64)
    if name == "soft_l1":
        return 1. / np.sqrt(1. + z)
    if name == "huber":
        g = 1.345
        with np.errstate(divide="ignore"):
            return np.where(z < g * g, 1., g / np.sqrt(z))
    if name == "cauchy":
        return 1. / (1. + z)
    if name == "arctan":
        return 1. / (1. + z ** 2)
    raise ValueError(name)


# --------------------------------------------------------------------------
# Primal-dual step-size schedules (reference primal_dual_solver.py:278-403)
# --------------------------------------------------------------------------
def pd_schedule(alg_type, L2, lmbda, iterations):
    """Returns arrays (sigma_n, tau_n, theta_n) of length `iterations`:
    sigma_n/tau_n are the values used INSIDE iteration n (before the update),
    theta_n the over-relaxation used at its end (after the update)."""
    L2 = float(L2)
    if alg_type == "ALG2":
        tau = 1. / np.sqrt(L2)
        sigma = 1. / (L2 * tau)
        gamma = 0.35 * lmbda
    elif alg_type == "ALG2_AHMOD":
        tau = 0.02
        sigma = 4. / (L2 * tau)
        gamma = 0.35 * lmbda
    elif alg_type == "ALG3":
        gamma_ = lmbda
        delta = 0.05
        mu = 2. * np.sqrt(gamma_ * delta / L2)
        theta_c = 1. / (1. + mu)
        sigma = mu / (2. * delta)
        tau = mu / (2. * gamma_)
    else:
        raise KeyError(alg_type)
    sig, ta, th = [], [], []
    for _ in range(iterations):
        sig.append(sigma)
        ta.append(tau)
        if alg_type == "ALG3":
            theta = theta_c
        else:
            theta = 1. / np.sqrt(1. + 2. * gamma * tau)
            tau = tau * theta
            sigma = sigma / theta
            if alg_type == "ALG2_AHMOD":
                theta = 0.
        th.append(theta)
    return np.array(sig), np.array(ta), np.array(th)


def primal_dual_denoise(b, shape, reg="TV", data="L2", alpha=0.03,
                        iterations=10, L2=8., alg_type="ALG2", x_scale=None,
                        spacing=None, x0=None):
    """Chambolle-Pock loop of primal_dual_solver.py:215-263 wired as
    run_denoising.py:95-154 (x0 = b, x_scale = max(b), unit spacing by
    default).  `b` is flat; returns the flat reconstruction (x * x_scale)."""
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    x_scale = float(np.max(b)) if x_scale is None else float(x_scale)
    x0 = b if x0 is None else np.asarray(x0, np.float64).reshape(-1)
    d = len(shape)
    lmbda = 1. / float(alpha)
    sig, ta, th = pd_schedule(alg_type, L2, lmbda, iterations)
    x = x0 / x_scale
    xbar = x.copy()
    p = 0
    for n in range(iterations):
        q = p + sig[n] * grad(xbar.reshape(shape), spacing).reshape(-1)
        p = prox_huber_conj(q, sig[n]) if reg == "Huber" else \
            prox_tv_conj(q, sig[n])
        Zshape = (d * shape[0],) + tuple(shape[1:]) if d > 1 else shape
        u = x - ta[n] * grad_adj(p.reshape(Zshape), spacing).reshape(-1)
        if data == "L2":
            xn = prox_ell2_denoising(u, ta[n] * lmbda, b, x_scale)
        else:
            xn = prox_ell1_denoising(u, ta[n] * lmbda, b, x_scale)
        xbar = xn + th[n] * (xn - x)
        x = xn
    return x * x_scale


# --------------------------------------------------------------------------
# LSMR (Fong & Saunders 2011) as called by the reference:
# scipy.sparse.linalg.lsmr(A, b, maxiter=iter_max, atol=0, btol=0), damp=0,
# conlim=1e8, no x0  (reference tikhonov_linear_solver.py:146-154).
# --------------------------------------------------------------------------
def _sym_ortho(a, b):
    """Stable Givens rotation (c, s, r) with r = hypot(a, b), following the
    LSQR/LSMR papers' SymOrtho."""
    if b == 0:
        return np.sign(a), 0, abs(a)
    if a == 0:
        return 0, np.sign(b), abs(b)
    if abs(b) > abs(a):
        t = a / b
        s = np.sign(b) / np.sqrt(1 + t * t)
        c = s * t
        r = b / s
    else:
        t = b / a
        c = np.sign(a) / np.sqrt(1 + t * t)
        s = c * t
        r = a / c
    return c, s, r


def lsmr(matvec, rmatvec, b, n, maxiter, atol=0., btol=0., conlim=1e8):
    """Returns (x, istop, itn)."""
    b = np.asarray(b, dtype=np.float64)
    norm = np.linalg.norm
    u = b.copy()
    normb = norm(b)
    x = np.zeros(n)
    beta = normb
    if beta > 0:
        u = (1 / beta) * u
        v = rmatvec(u)
        alpha = norm(v)
    else:
        v = np.zeros(n)
        alpha = 0
    if alpha > 0:
        v = (1 / alpha) * v

    itn = 0
    zetabar = alpha * beta
    alphabar = alpha
    rho = rhobar = cbar = 1
    sbar = 0
    h = v.copy()
    hbar = np.zeros(n)
    betadd = beta
    betad = 0
    rhodold = 1
    tautildeold = 0
    thetatilde = 0
    zeta = 0
    d = 0
    normA2 = alpha * alpha
    maxrbar = 0
    minrbar = 1e+100
    istop = 0
    ctol = 1 / conlim if conlim > 0 else 0
    if alpha * beta == 0:
        return x, istop, itn
    if normb == 0:
        x[()] = 0
        return x, istop, itn

    while itn < maxiter:
        itn += 1
        u *= -alpha
        u += matvec(v)
        beta = norm(u)
        if beta > 0:
            u *= (1 / beta)
            v *= -beta
            v += rmatvec(u)
            alpha = norm(v)
            if alpha > 0:
                v *= (1 / alpha)

        chat, shat, alphahat = _sym_ortho(alphabar, 0.)
        rhoold = rho
        c, s, rho = _sym_ortho(alphahat, beta)
        thetanew = s * alpha
        alphabar = c * alpha

        rhobarold = rhobar
        zetaold = zeta
        thetabar = sbar * rho
        rhotemp = cbar * rho
        cbar, sbar, rhobar = _sym_ortho(cbar * rho, thetanew)
        zeta = cbar * zetabar
        zetabar = -sbar * zetabar

        hbar *= -(thetabar * rho / (rhoold * rhobarold))
        hbar += h
        x += (zeta / (rho * rhobar)) * hbar
        h *= -(thetanew / rho)
        h += v

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
        normr = np.sqrt(d + (betad - taud) ** 2 + betadd * betadd)

        normA2 = normA2 + beta * beta
        normA = np.sqrt(normA2)
        normA2 = normA2 + alpha * alpha
        maxrbar = max(maxrbar, rhobarold)
        if itn > 1:
            minrbar = min(minrbar, rhobarold)
        condA = max(maxrbar, rhotemp) / min(minrbar, rhotemp)

        normar = abs(zetabar)
        normx = norm(x)
        test1 = normr / normb
        test2 = normar / (normA * normr) if (normA * normr) != 0 else np.inf
        test3 = 1 / condA
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


# --------------------------------------------------------------------------
# Tikhonov solver (reference tikhonov_linear_solver.py:120-280)
# --------------------------------------------------------------------------
def tikhonov(A, A_adj, B, B_adj, b, x0, alpha=0.01, b_reg=0.,
             data_loss="linear", data_loss_scale=1., minimizer="lsmr",
             iter_max=10, x_scale=1., bounds=(0, np.inf)):
    """Flat-array callables A, A_adj, B, B_adj (as the reference takes them).
    Returns the flat solution multiplied by x_scale."""
    if minimizer == "lsmr" and data_loss != "linear":
        raise ValueError("lsmr solver cannot be used with non-linear data loss")
    x_scale = float(x_scale)
    x0 = np.array(x0, dtype=np.float64) / x_scale
    b = np.asarray(b, np.float64) / x_scale
    b_reg = b_reg / x_scale
    alpha = float(alpha)
    if bounds is not None:
        x0 = np.clip(x0, bounds[0], bounds[1])

    if minimizer == "lsmr":
        if alpha > EPS:
            sa = np.sqrt(alpha)
            nb = b.size
            mv = lambda x: np.concatenate((A(x), sa * B(x)))
            rmv = lambda y: A_adj(y[:nb]) + sa * B_adj(y[nb:])
            rhs = np.zeros(mv(x0).size)
            rhs[:nb] = b
            rhs[nb:] = sa * b_reg
        else:
            mv, rmv, rhs = A, A_adj, b
        x = lsmr(mv, rmv, rhs, x0.size, iter_max)[0]
        if bounds is not None:
            x = np.clip(x, bounds[0], bounds[1])
        return x * x_scale

    import scipy.optimize
    import scipy.sparse.linalg
    if minimizer in ("lsq_linear", "least_squares"):
        # SciPy drivers over the augmented operator, called as the reference
        # does (tikhonov_linear_solver.py:160-195)
        if minimizer == "lsq_linear" and data_loss != "linear":
            raise ValueError(
                "lsq_linear solver cannot be used with non-linear data loss")
        if alpha > EPS:
            sa = np.sqrt(alpha)
            nb = b.size
            mv = lambda x: np.concatenate((A(x), sa * B(x)))
            rmv = lambda y: A_adj(y[:nb]) + sa * B_adj(y[nb:])
            rhs = np.zeros(mv(x0).size)
            rhs[:nb] = b
            rhs[nb:] = sa * b_reg
        else:
            mv, rmv, rhs = A, A_adj, b
        op = scipy.sparse.linalg.LinearOperator(
            shape=(rhs.size, x0.size), matvec=mv, rmatvec=rmv)
        if minimizer == "lsq_linear":
            x = scipy.optimize.lsq_linear(
                op, rhs, max_iter=iter_max, lsq_solver='lsmr',
                lsmr_tol='auto', bounds=bounds).x
        else:
            x = scipy.optimize.least_squares(
                fun=lambda x: op * x - rhs, jac=lambda x: op,
                jac_sparsity=lambda x: op, x0=x0, tr_solver='lsmr',
                bounds=bounds, loss=data_loss, f_scale=data_loss_scale,
                max_nfev=iter_max).x
        return x * x_scale

    # robust-loss branch: scipy.optimize.minimize (third-party, the reference
    # calls it the same way, tikhonov_linear_solver.py:197-220).  b_reg is
    # ignored by the reference here (regulariser 1/2||Bx||^2).

    def cost_data(x):
        r = A(x) - b
        return 0.5 * np.sum(loss(data_loss, r ** 2, data_loss_scale))

    def grad_data(x):
        r = A(x) - b
        return A_adj(gradient_loss(data_loss, r ** 2, data_loss_scale) * r)

    if alpha > EPS:
        cost = lambda x: cost_data(x) + alpha * 0.5 * np.sum(B(x) ** 2)
        jac = lambda x: grad_data(x) + alpha * B_adj(B(x))
    else:
        cost, jac = cost_data, grad_data
    bnds = [[bounds[0], bounds[1]]] * x0.size
    x = scipy.optimize.minimize(method=minimizer, fun=cost, jac=jac, x0=x0,
                                bounds=bnds,
                                options={'maxiter': iter_max, 'disp': 0}).x
    return x * x_scale


# --------------------------------------------------------------------------
# ADMM (reference admm_linear_solver.py:165-309)
# --------------------------------------------------------------------------
def admm_prox_g(t, tau, dimension):
    """Isotropic vector soft-threshold, admm_linear_solver.py:239-253."""
    parts = np.array_split(t, dimension)
    nrm2 = parts[0] ** 2
    for i in range(1, dimension):
        nrm2 = nrm2 + parts[i] ** 2
    nrm = np.sqrt(nrm2)
    ind = nrm > tau
    v = np.zeros_like(t)
    m = parts[0].shape[0]
    for i in range(dimension):
        vt = v[i * m:(i + 1) * m, ...]
        vt[ind] = np.maximum(np.abs(nrm[ind]) - tau, 0) * np.sign(nrm[ind]) \
            * parts[i][ind] / nrm[ind]
    return v


def admm(A, A_adj, B, B_adj, b, x0, dimension, b_reg=0., alpha=0.01,
         iter_max=10, minimizer="lsmr", data_loss="linear", rho=0.5,
         iterations=10, x_scale=1.):
    """ADMM loop; flat-array callables.  Returns x * x_scale."""
    x_scale = float(x_scale)
    x = np.array(x0, dtype=np.float64) / x_scale
    bs = np.asarray(b, np.float64) / x_scale
    c = b_reg / x_scale
    v = B(x) - c
    w = np.zeros_like(v)
    for _ in range(iterations):
        x = tikhonov(A, A_adj, B, B_adj, bs, x, alpha=rho, b_reg=v - w + c,
                     data_loss=data_loss, minimizer=minimizer,
                     iter_max=iter_max, x_scale=1.)
        t = B(x) + w - c
        v = admm_prox_g(t, alpha / float(rho), dimension)
        w = t - v
    return x * x_scale


# --------------------------------------------------------------------------
# Generic primal-dual loop with caller-supplied proxes (PD deconvolution:
# prox_f = prox_linear_least_squares, interface :257-280, solvers_test.py:146-158)
# --------------------------------------------------------------------------
def prox_linear_least_squares(x, tau, A, A_adj, b, x0, iter_max=10,
                              x_scale=1., bounds=(0, np.inf)):
    """proximal_operators.py:43-78: Tikhonov solve with B = I, b_reg = x,
    alpha = 1/tau.  As in the reference, data and start are divided by x_scale
    here AND x_scale is handed on to the solver, which divides b, x0 and b_reg
    once more and multiplies its result by x_scale (:62-78)."""
    ident = lambda v: v.reshape(-1)
    x_scale = float(x_scale)
    return tikhonov(A, A_adj, ident, ident, np.asarray(b) / x_scale,
                    np.asarray(x0) / x_scale, alpha=1. / tau, b_reg=x,
                    iter_max=iter_max, x_scale=x_scale, bounds=bounds)


def primal_dual(prox_f, prox_g_conj, B, B_conj, L2, x0, alpha=0.01,
                iterations=10, x_scale=1., alg_type="ALG2"):
    """primal_dual_solver.py:215-263 with flat callables; returns x * x_scale."""
    x_scale = float(x_scale)
    lmbda = 1. / float(alpha)
    sig, ta, th = pd_schedule(alg_type, L2, lmbda, iterations)
    x = np.array(x0, dtype=np.float64) / x_scale
    xbar = x.copy()
    p = 0
    for n in range(iterations):
        p = prox_g_conj(p + sig[n] * B(xbar), sig[n])
        xn = prox_f(x - ta[n] * B_conj(p), ta[n] * lmbda)
        xbar = xn + th[n] * (xn - x)
        x = xn
    return x * x_scale


# --------------------------------------------------------------------------
# Regulariser values and similarity measures (observer side)
# --------------------------------------------------------------------------
def _sum_sq_split(Dx, dimension):
    parts = np.array_split(Dx, dimension)
    acc = parts[0] ** 2
    for i in range(1, len(parts)):
        acc = acc + parts[i] ** 2
    return acc


def prior_tk0(x):
    return 0.5 * np.sum(np.square(x))                 # prior_measures.py:19-21


def prior_tk1(x, D):
    return 0.5 * np.sum(np.square(D(x)))              # prior_measures.py:23-25


def prior_tv(x, D, dimension):
    return np.sum(np.sqrt(_sum_sq_split(D(x), dimension)))   # :27-38


def prior_huber(x, D, dimension, gamma=0.05):
    # prior_measures.py:40-52 -> LossFunctions.huber(f2, gamma),
    # loss_functions.py:148-158 with f_scale = 1: f2 below gamma^2 stays,
    # above it 2*gamma*sqrt(f2) - gamma^2; divided by 2*gamma
    f2 = _sum_sq_split(D(x), dimension)
    h = np.where(f2 < gamma * gamma, f2, 2. * gamma * np.sqrt(f2) - gamma ** 2)
    return np.sum(h / (2. * gamma))


def sim_sad(x, x_ref):                                # similarity_measures.py:26-29
    return np.sum(np.abs(x - x_ref))


def sim_mae(x, x_ref):                                # :40-44
    return sim_sad(x, x_ref) / float(x.size)


def sim_ssd(x, x_ref):                                # :55-59
    return np.sum(np.square(x - x_ref))


def sim_mse(x, x_ref):                                # :70-74
    return sim_ssd(x, x_ref) / float(x.size)


def sim_rmse(x, x_ref):                               # :85-87
    return np.sqrt(sim_mse(x, x_ref))


def sim_psnr(x, x_ref):                               # :98-101 (unguarded mse = 0)
    with np.errstate(divide="ignore"):
        return 10 * np.log10(np.max(x_ref) ** 2 / sim_mse(x, x_ref))


def sim_ncc(x, x_ref):                                # :112-120
    ncc = np.sum((x - x.mean()) * (x_ref - x_ref.mean()))
    return ncc / float(x.size * x.std(ddof=1) * x_ref.std(ddof=1))


# --------------------------------------------------------------------------
# Flat-callable factories (mirror how run_denoising.py:104-107 and
# run_deconvolution.py:120-129 wrap the N-D operators)
# --------------------------------------------------------------------------
def flat_operators(shape, spacing=None, cov=None, alpha_cut=3):
    d = len(shape)
    Z = (d * shape[0],) + tuple(shape[1:]) if d > 1 else tuple(shape)
    D = lambda x: grad(x.reshape(shape), spacing).reshape(-1)
    D_adj = lambda p: grad_adj(p.reshape(Z), spacing).reshape(-1)
    if cov is None:
        return D, D_adj
    taps = gaussian_taps(d, cov, spacing, alpha_cut)
    A = lambda x: convolve_nd(x.reshape(shape), taps, "wrap").reshape(-1)
    return D, D_adj, A, A


# --------------------------------------------------------------------------
# Synthetic inputs (build-owned, not from the reference)
# --------------------------------------------------------------------------
def synth_volume(n, seed=0, kind="gauss", dtype=np.float64):
    q = max(n // 4, 1)
    i = np.arange(n)
    blk = (i // q)
    chk = (blk[:, None, None] + blk[None, :, None] + blk[None, None, :]) % 2
    v = 100.0 * chk.astype(np.float64)
    c = n / 2.0
    r2 = ((i - c) ** 2)
    ball = (r2[:, None, None] + r2[None, :, None] + r2[None, None, :]) \
        < (n / 3.0) ** 2
    v += 50.0 * ball
    rng = np.random.default_rng(seed)
    if kind == "gauss":
        v = v + 0.05 * v.max() * rng.standard_normal(v.shape)
    elif kind == "sp":
        u = rng.random(v.shape)
        v = np.where(u < 0.05, 0.0, np.where(u > 0.95, 150.0, v))
    elif kind != "clean":
        raise ValueError(kind)
    return v.astype(dtype)


# --------------------------------------------------------------------------
# Reference-STYLE iteration for CPU timing (same work as the reference does:
# ndimage correlate calls, concatenate, flatten copies, temporaries)
# --------------------------------------------------------------------------
def pd_tvl2_refstyle(b, shape, alpha, iterations, L2, x_scale):
    import scipy.ndimage
    d = len(shape)
    h = 1.0
    kf, kb = [], []
    for a in range(d):
        shp_f = [1] * d
        shp_b = [1] * d
        shp_f[d - 1 - a] = 2
        shp_b[d - 1 - a] = 3
        kf.append((np.array([1., -1.]) / h).reshape(shp_f))
        kb.append((-np.array([0., 1., -1.]) / h).reshape(shp_b))
    conv = scipy.ndimage.convolve
    Zs = (d * shape[0],) + tuple(shape[1:])

    def D(x):
        x = x.reshape(*shape)
        return np.concatenate(
            [conv(x, kf[a], mode="constant") for a in range(d)]).flatten()

    def D_adj(p):
        parts = np.array_split(p.reshape(*Zs), d)
        out = conv(parts[0], kb[0], mode="constant")
        for a in range(1, d):
            out += conv(parts[a], kb[a], mode="constant")
        return out.flatten()

    lmbda = 1. / alpha
    sig, ta, th = pd_schedule("ALG2", L2, lmbda, iterations)
    x_n = np.array(b / x_scale)
    x_mean = np.array(x_n)
    p_n = 0
    for n in range(iterations):
        p_n = prox_tv_conj(p_n + sig[n] * D(x_mean), sig[n])
        x_np1 = prox_ell2_denoising(x_n - ta[n] * D_adj(p_n), ta[n] * lmbda,
                                    b, x_scale)
        x_mean = x_np1 + th[n] * (x_np1 - x_n)
        x_n = x_np1
    return x_n * x_scale


def admm_lsmr_refstyle(y, shape, cov, alpha, rho, iterations, iter_max,
                       x_scale):
    """ADMM deconvolution (BASELINE config 4, lsmr branch) doing the work the
    reference does: scipy.ndimage.convolve with the DENSE Gaussian taps
    (linear_operators.py:60-86), ndimage differences + concatenate
    (:121-169), scipy.sparse.linalg.lsmr on the augmented LinearOperator
    (tikhonov_linear_solver.py:146-158, :226-274), the outer loop of
    admm_linear_solver.py:165-253.  Timing stand-in for the reference on the
    GPU box; held to admm() in tests/test_oracle_golden.py."""
    import scipy.ndimage
    import scipy.sparse.linalg
    d = len(shape)
    taps = gaussian_taps(d, cov)
    kf, kb = [], []
    for a in range(d):
        shp_f, shp_b = [1] * d, [1] * d
        shp_f[d - 1 - a] = 2
        shp_b[d - 1 - a] = 3
        kf.append(np.array([1., -1.]).reshape(shp_f))
        kb.append((-np.array([0., 1., -1.])).reshape(shp_b))
    conv = scipy.ndimage.convolve
    Zs = (d * shape[0],) + tuple(shape[1:]) if d > 1 else tuple(shape)
    A = lambda x: conv(x.reshape(*shape), taps, mode="wrap").flatten()

    def D(x):
        x = x.reshape(*shape)
        parts = [conv(x, kf[a], mode="constant") for a in range(d)]
        return (np.concatenate(parts) if d > 1 else parts[0]).flatten()

    def D_adj(p):
        parts = np.array_split(p.reshape(*Zs), d)
        out = conv(parts[0], kb[0], mode="constant")
        for a in range(1, d):
            out += conv(parts[a], kb[a], mode="constant")
        return out.flatten()

    x_scale = float(x_scale)
    b = np.asarray(y, np.float64) / x_scale
    x = np.array(b)
    v = D(x)
    w = np.zeros_like(v)
    sr = np.sqrt(rho)
    nb = b.size
    for _ in range(iterations):
        breg = v - w
        x0 = np.clip(x, 0, np.inf)
        fw = lambda z: np.concatenate((A(z), sr * D(z)))
        bw = lambda z: A(z[:nb]) + sr * D_adj(z[nb:])
        rhs = np.zeros(fw(x0).size)
        rhs[:nb] = b
        rhs[nb:] = sr * breg
        op = scipy.sparse.linalg.LinearOperator(
            shape=(rhs.size, x0.size), matvec=fw, rmatvec=bw)
        x = scipy.sparse.linalg.lsmr(op, rhs, maxiter=iter_max, atol=0,
                                     btol=0)[0]
        x = np.clip(x, 0, np.inf)
        t = D(x) + w
        v = admm_prox_g(t.reshape(Zs), alpha / float(rho), d).reshape(-1)
        w = t - v
    return x * x_scale
