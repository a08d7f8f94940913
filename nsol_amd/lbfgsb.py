"""L-BFGS-B (Byrd, Lu, Nocedal, Zhu, SIAM J. Sci. Comput. 16, 1995; subspace
step of Morales & Nocedal, ACM TOMS 38, 2011; line search of More & Thuente,
ACM TOMS 20, 1994) with every length-n vector resident in HBM.

Replaces the host driver behind
    scipy.optimize.minimize(method="L-BFGS-B", fun, jac, x0, bounds=[[lo, hi]]*n,
                            options={'maxiter': iter_max})
of tikhonov_linear_solver.py:197-220 for the robust-loss branch, with SciPy's
defaults (maxcor = 10, ftol = 2.22e-9, gtol = 1e-5, maxls = 20).  SciPy's own
driver keeps all O(n) work (Cauchy-point search, subspace step, updates of the
limited-memory matrices) in single-threaded host code: ~15 s per iteration at
n = 256^3 even for a trivial objective.

Structure: this module holds the iteration logic and the small (2m x 2m)
dense algebra on the host in float64; all length-n work goes through a
`backend` (nsol_amd.lbfgsb_device.DeviceBackend in the product; the tests
also drive the same logic with a NumPy backend to compare it with SciPy on the
CPU).  Bounds are uniform (lo, hi) as in the reference (`bounds=(0, inf)`).
"""
import math

import numpy as np

# counters of the last minimize() call (diagnostics for tools/profile_admm.py)
STATS = {"cauchy_calls": 0, "breakpoints": 0, "crossed": 0, "fetches": 0}

EPSMCH = np.finfo(np.float64).eps
# False: direction, projection, d = z - x and the BFGS update's products with d as
# separate passes (the A/B reference of the tests) even when the backend offers
# them fused
FUSE_SUBSPACE_STEP = True
# True: scalars of kernels that follow each other without a decision in between are
# read back together where the backend offers it (cauchy_setup + W'd; the count of
# free variables + the subspace matrix)
MERGE_READBACKS = True
# True: where the backend's subspace step can form the reduced gradient r itself
# (subspace_step_forms_r), the Gram pass neither forms nor stores it and the step does
# not read it; r becomes an array only on the rare path that needs one
DEFER_REDUCED_GRADIENT = True
# False: [Y S]'Z r by a pass of its own even when the Gram pass delivered it
USE_GRAM_RHS = True
BIG = 1.0e10
FTOL, GTOL, XTOL = 1.0e-3, 0.9, 0.1     # line search constants of L-BFGS-B


# ---------------------------------------------------------------------------
# More-Thuente line search (dcsrch / dcstep)
# ---------------------------------------------------------------------------
class LineSearch(object):
    """Reverse-communication search for a step satisfying the strong Wolfe
    conditions  f(stp) <= f(0) + ftol*stp*f'(0),  |f'(stp)| <= gtol*|f'(0)|."""

    def __init__(self, f, g, stp, stpmin, stpmax):
        self.stpmin, self.stpmax = stpmin, stpmax
        self.task = "FG"
        if stp < stpmin:
            self.task = "ERROR: STP .LT. STPMIN"
        if stp > stpmax:
            self.task = "ERROR: STP .GT. STPMAX"
        if g >= 0:
            self.task = "ERROR: INITIAL G .GE. ZERO"
        self.brackt = False
        self.stage = 1
        self.finit, self.ginit = f, g
        self.gtest = FTOL * g
        self.width = stpmax - stpmin
        self.width1 = self.width / 0.5
        self.stx, self.fx, self.gx = 0.0, f, g
        self.sty, self.fy, self.gy = 0.0, f, g
        self.stmin = 0.0
        self.stmax = stp + 4.0 * stp
        self.stp = stp

    def step(self, f, g):
        """Feed f, f' at self.stp; updates self.stp / self.task."""
        stp = self.stp
        ftest = self.finit + stp * self.gtest
        if self.stage == 1 and f <= ftest and g >= 0:
            self.stage = 2
        task = "FG"
        if self.brackt and (stp <= self.stmin or stp >= self.stmax):
            task = "WARNING: ROUNDING ERRORS PREVENT PROGRESS"
        if self.brackt and self.stmax - self.stmin <= XTOL * self.stmax:
            task = "WARNING: XTOL TEST SATISFIED"
        if stp == self.stpmax and f <= ftest and g <= self.gtest:
            task = "WARNING: STP = STPMAX"
        if stp == self.stpmin and (f > ftest or g >= self.gtest):
            task = "WARNING: STP = STPMIN"
        if f <= ftest and abs(g) <= GTOL * (-self.ginit):
            task = "CONVERGENCE"
        if task[:4] in ("WARN", "CONV"):
            self.task = task
            return
        if self.stage == 1 and f <= self.fx and f > ftest:
            # modified function psi(stp) = f(stp) - f(0) - stp*gtest
            fm = f - stp * self.gtest
            fxm = self.fx - self.stx * self.gtest
            fym = self.fy - self.sty * self.gtest
            gm = g - self.gtest
            gxm = self.gx - self.gtest
            gym = self.gy - self.gtest
            (self.stx, fxm, gxm, self.sty, fym, gym, stp,
             self.brackt) = _dcstep(self.stx, fxm, gxm, self.sty, fym, gym,
                                    stp, fm, gm, self.brackt, self.stmin,
                                    self.stmax)
            self.fx = fxm + self.stx * self.gtest
            self.fy = fym + self.sty * self.gtest
            self.gx = gxm + self.gtest
            self.gy = gym + self.gtest
        else:
            (self.stx, self.fx, self.gx, self.sty, self.fy, self.gy, stp,
             self.brackt) = _dcstep(self.stx, self.fx, self.gx, self.sty,
                                    self.fy, self.gy, stp, f, g, self.brackt,
                                    self.stmin, self.stmax)
        if self.brackt:
            if abs(self.sty - self.stx) >= 0.66 * self.width1:
                stp = self.stx + 0.5 * (self.sty - self.stx)
            self.width1 = self.width
            self.width = abs(self.sty - self.stx)
        if self.brackt:
            self.stmin = min(self.stx, self.sty)
            self.stmax = max(self.stx, self.sty)
        else:
            self.stmin = stp + 1.1 * (stp - self.stx)
            self.stmax = stp + 4.0 * (stp - self.stx)
        stp = max(stp, self.stpmin)
        stp = min(stp, self.stpmax)
        if (self.brackt and (stp <= self.stmin or stp >= self.stmax)) or \
                (self.brackt and self.stmax - self.stmin <= XTOL * self.stmax):
            stp = self.stx
        self.stp = stp
        self.task = "FG"


def _dcstep(stx, fx, dx, sty, fy, dy, stp, fp, dp, brackt, stpmin, stpmax):
    """Safeguarded cubic / quadratic step of More-Thuente."""
    sgnd = dp * (dx / abs(dx))
    if fp > fx:                                   # case 1: higher value
        theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp
        s = max(abs(theta), abs(dx), abs(dp))
        gamma = s * math.sqrt((theta / s) ** 2 - (dx / s) * (dp / s))
        if stp < stx:
            gamma = -gamma
        p = (gamma - dx) + theta
        q = ((gamma - dx) + gamma) + dp
        r = p / q
        stpc = stx + r * (stp - stx)
        stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * \
            (stp - stx)
        if abs(stpc - stx) < abs(stpq - stx):
            stpf = stpc
        else:
            stpf = stpc + (stpq - stpc) / 2.0
        brackt = True
    elif sgnd < 0.0:                              # case 2: derivative flips
        theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp
        s = max(abs(theta), abs(dx), abs(dp))
        gamma = s * math.sqrt((theta / s) ** 2 - (dx / s) * (dp / s))
        if stp > stx:
            gamma = -gamma
        p = (gamma - dp) + theta
        q = ((gamma - dp) + gamma) + dx
        r = p / q
        stpc = stp + r * (stx - stp)
        stpq = stp + (dp / (dp - dx)) * (stx - stp)
        stpf = stpc if abs(stpc - stp) > abs(stpq - stp) else stpq
        brackt = True
    elif abs(dp) < abs(dx):                       # case 3: derivative shrinks
        theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp
        s = max(abs(theta), abs(dx), abs(dp))
        gamma = s * math.sqrt(max(0.0, (theta / s) ** 2 - (dx / s) * (dp / s)))
        if stp > stx:
            gamma = -gamma
        p = (gamma - dp) + theta
        q = (gamma + (dx - dp)) + gamma
        r = p / q
        if r < 0.0 and gamma != 0.0:
            stpc = stp + r * (stx - stp)
        elif stp > stx:
            stpc = stpmax
        else:
            stpc = stpmin
        stpq = stp + (dp / (dp - dx)) * (stx - stp)
        if brackt:
            stpf = stpc if abs(stpc - stp) < abs(stpq - stp) else stpq
            if stp > stx:
                stpf = min(stp + 0.66 * (sty - stp), stpf)
            else:
                stpf = max(stp + 0.66 * (sty - stp), stpf)
        else:
            stpf = stpc if abs(stpc - stp) > abs(stpq - stp) else stpq
            stpf = min(stpmax, stpf)
            stpf = max(stpmin, stpf)
    else:                                         # case 4
        if brackt:
            theta = 3.0 * (fp - fy) / (sty - stp) + dy + dp
            s = max(abs(theta), abs(dy), abs(dp))
            gamma = s * math.sqrt((theta / s) ** 2 - (dy / s) * (dp / s))
            if stp > sty:
                gamma = -gamma
            p = (gamma - dp) + theta
            q = ((gamma - dp) + gamma) + dy
            r = p / q
            stpf = stp + r * (sty - stp)
        elif stp > stx:
            stpf = stpmax
        else:
            stpf = stpmin
    if fp > fx:
        sty, fy, dy = stp, fp, dp
    else:
        if sgnd < 0.0:
            sty, fy, dy = stx, fx, dx
        stx, fx, dx = stp, fp, dp
    return stx, fx, dx, sty, fy, dy, stpf, brackt


# ---------------------------------------------------------------------------
# Small dense algebra of the compact representation (host, float64)
# ---------------------------------------------------------------------------
class CompactMatrix(object):
    """B = theta*I - W M W^T with W = [Y, theta*S]; keeps S^T S (upper), the
    lower triangle of S^T Y and the Cholesky factor of
    T = theta*S^T S + L D^-1 L^T."""

    def __init__(self, m):
        self.m = m
        self.col = 0
        self.theta = 1.0
        self.ss = np.zeros((m, m))
        self.sy = np.zeros((m, m))
        self.wt = None

    def reset(self):
        self.col = 0
        self.theta = 1.0

    def form_t(self):
        # T = theta * S'S + L D^-1 L'  (L: strictly lower part of S'Y, D its diagonal);
        # the sums over the stored pairs as matrix products: interpreter loops over
        # col^3 / 3 terms held the GPU idle for a tenth of a millisecond per iteration
        c, th = self.col, self.theta
        sy = self.sy[:c, :c]
        ls = np.tril(sy, -1)
        ss = np.triu(self.ss[:c, :c])
        ss = ss + np.triu(ss, 1).T
        T = th * ss + (ls / np.diag(sy)[None, :]).dot(ls.T)
        T = np.triu(T) + np.triu(T, 1).T
        try:
            self.wt = np.linalg.cholesky(T).T        # upper: T = wt^T wt
        except np.linalg.LinAlgError:
            return False
        return True

    def bmv(self, v):
        """Product of the 2col x 2col middle matrix M with v."""
        c = self.col
        if c == 0:
            return np.zeros(0)
        sy, wt = self.sy[:c, :c], self.wt
        v = np.asarray(v, dtype=np.float64)
        dg = np.diag(sy)
        ls = np.tril(sy, -1)
        p2 = v[c:] + ls.dot(v[:c] / dg)
        p2 = _solve_upper_t(wt, p2)                  # wt^T z = rhs
        p2 = _solve_upper(wt, p2)                    # wt z = rhs
        p1 = -(v[:c] / np.sqrt(dg)) / np.sqrt(dg)
        p1 = p1 + ls.T.dot(p2) / dg
        return np.concatenate((p1, p2))


def middle_matrix(cm):
    """The 2col x 2col middle matrix M itself (bmv column by column, with the loops
    over the stored pairs as matrix products: the Cauchy search wants all of M once
    per iteration, and 2col calls of bmv were a millisecond of interpreter time)."""
    c = cm.col
    sy, wt = cm.sy[:c, :c], cm.wt
    dg = np.diag(sy).copy()
    ls = np.tril(sy, -1)
    v = np.eye(2 * c)
    p2 = v[c:] + ls.dot(v[:c] / dg[:, None])
    p2 = _solve_upper_t(wt, p2)
    p2 = _solve_upper(wt, p2)
    p1 = -(v[:c] / np.sqrt(dg)[:, None]) / np.sqrt(dg)[:, None]
    p1 = p1 + ls.T.dot(p2) / dg[:, None]
    return np.vstack((p1, p2))


def _solve_upper(R, b):
    from scipy.linalg import solve_triangular
    return solve_triangular(R, b, lower=False, check_finite=False)


def _solve_upper_t(R, b):
    from scipy.linalg import solve_triangular
    return solve_triangular(R, b, lower=False, trans='T', check_finite=False)


def form_k(cm, yzzy, szzs, szzy):
    """LEL^T factorisation of the indefinite subspace matrix; the masked Gram
    blocks are over the FREE variables: yzzy = Y^T Z Z^T Y, szzs = S^T Z Z^T S,
    szzy = S^T Z Z^T Y.  Returns (R11, W12, R22) or None."""
    c, th = cm.col, cm.theta
    saas = (np.triu(cm.ss[:c, :c]) + np.triu(cm.ss[:c, :c], 1).T) - szzs
    # (1,1) block  D + Y'ZZ'Y/theta
    k11 = yzzy / th + np.diag(np.diag(cm.sy[:c, :c]))
    # (1,2) block  -L_a' + R_z'  (row: y index, column: s index): above the
    # diagonal the strictly lower part of S'AA'Y = S'Y - S'ZZ'Y, negated and
    # transposed; on and below it R_z (the upper part of S'ZZ'Y incl. its diagonal)
    k12 = np.triu(szzy).T - np.tril(cm.sy[:c, :c] - szzy, -1).T
    try:
        r11 = np.linalg.cholesky(k11).T               # k11 = r11^T r11
    except np.linalg.LinAlgError:
        return None
    w12 = _solve_upper_t(r11, k12)                    # r11^-T k12
    k22 = th * saas + w12.T.dot(w12)
    try:
        r22 = np.linalg.cholesky(k22).T
    except np.linalg.LinAlgError:
        return None
    return r11, w12, r22


def solve_k(fac, wv, c):
    """wv := K^-1 wv with the factors of form_k."""
    r11, w12, r22 = fac
    out = np.array(wv, dtype=np.float64)
    # forward: [r11^T 0; w12^T r22^T] z = wv
    z1 = _solve_upper_t(r11, out[:c])
    z2 = _solve_upper_t(r22, out[c:] - w12.T.dot(z1))
    z1 = -z1
    # backward: [r11 w12; 0 r22] x = z
    x2 = _solve_upper(r22, z2)
    x1 = _solve_upper(r11, z1 - w12.dot(x2))
    out[:c], out[c:] = x1, x2
    return out


# ---------------------------------------------------------------------------
# Driver
# ---------------------------------------------------------------------------
def minimize(fun_and_grad, x0, lo, hi, backend, maxiter=10, m=10,
             factr=1.0e7, pgtol=1.0e-5, maxls=20, maxfun=15000, start=None):
    """Minimise f subject to lo <= x <= hi (uniform bounds, +-inf allowed).

    fun_and_grad(x) -> (float f, vector g) with backend vectors.  Returns
    (x, info dict).  Follows scipy's iteration accounting: stops after
    `maxiter` accepted iterations.

    start: (f, g[, |proj g|_inf]) at x0 when the caller already holds them (x0
    feasible, so that the clip below returns its values: e.g. the point an earlier
    call on the same objective returned, with info["fun"] / info["jac"] /
    info["pg"]); the first evaluation is then not repeated (it still counts in
    nfev, as scipy would report it).

    fun_and_grad.with_extras(x, d, lo, hi, gold) -> (f, g, g'd or None, |proj g|_inf,
    g - gold or None, its squared norm or None), when the objective offers it, replaces
    the evaluation, backend.dot(g, d), backend.projgr(x, g, lo, hi) and
    backend.diff_dots(g, gold) of every new point (d, gold may be None)."""
    be = backend
    # A backend whose kernels want whole vectors (DeviceBackend.pad_to: 16
    # elements -- 16-byte accesses and one mask byte per element) gets the
    # problem padded with inert variables: x = clip(0), gradient 0.  They sit at
    # a bound (or are free with a zero gradient), never move, and add exact
    # zeros to every inner product, so the iterates of the first n variables are
    # those of the unpadded problem.
    pad_to = getattr(be, "pad_to", 0)
    n_in = be.size(x0)
    if pad_to and n_in % pad_to:
        n_pad = n_in + (-n_in) % pad_to

        def padded(xp):
            f, g = fun_and_grad(be.head(xp, n_in))
            return f, be.pad(g, n_pad)
        if start is not None:
            start = (start[0], be.pad(start[1], n_pad)) + tuple(start[2:])
        if getattr(fun_and_grad, "with_extras", None) is not None:
            def padded_extras(xp, dp, lo_, hi_, goldp=None):
                f, g, gd, pg, yd, rr = fun_and_grad.with_extras(
                    be.head(xp, n_in), None if dp is None else be.head(dp, n_in),
                    lo_, hi_, None if goldp is None else be.head(goldp, n_in))
                return (f, be.pad(g, n_pad), gd, pg,
                        None if yd is None else be.pad(yd, n_pad), rr)
            padded.with_extras = padded_extras
        xp, info = minimize(padded, be.pad(x0, n_pad), lo, hi, be,
                            maxiter=maxiter, m=m, factr=factr, pgtol=pgtol,
                            maxls=maxls, maxfun=maxfun, start=start)
        if info.get("jac") is not None:
            info["jac"] = be.copy(be.head(info["jac"], n_in))
        return be.copy(be.head(xp, n_in)), info
    has_lo, has_hi = np.isfinite(lo), np.isfinite(hi)
    cnstnd = has_lo or has_hi
    boxed = has_lo and has_hi
    x = be.clip(x0, lo, hi) if start is None else x0
    cm = CompactMatrix(m)
    ws, wy = [], []                   # stored s_k, y_k (oldest first)
    iwhere = be.init_where(x, lo, hi)
    ext = getattr(fun_and_grad, "with_extras", None)
    pg = None
    if start is not None:
        f, g = start[0], start[1]
        pg = start[2] if len(start) > 2 else None
    elif ext is not None:
        f, g, _, pg = ext(x, None, lo, hi)[:4]
    else:
        f, g = fun_and_grad(x)
    nfgv = 1
    it = 0
    nskip = 0
    updatd = False
    sbgnrm = be.projgr(x, g, lo, hi) if pg is None else pg
    info = {"task": "START", "nit": 0, "nfev": 1}
    if sbgnrm <= pgtol:
        info["task"] = "CONVERGENCE: NORM_OF_PROJECTED_GRADIENT_<=_PGTOL"
        info["fun"], info["jac"], info["pg"] = f, g, sbgnrm
        return x, info

    while True:
        col, theta = cm.col, cm.theta
        # ---------------- generalized Cauchy point ------------------------
        if not cnstnd and col > 0:
            z = be.copy(x)
            c_vec = np.zeros(2 * col)
            nfree_all = True
        else:
            z, c_vec, iwhere = _cauchy(be, x, g, lo, hi, iwhere, ws, wy, cm,
                                       sbgnrm)
            nfree_all = False
        # ---------------- subspace minimisation ---------------------------
        step = None
        # (with the fused Gram pass below the count is enqueued ahead of it and read
        # back with its result: the pass is wasted in the rare case of no free variable)
        late_count = MERGE_READBACKS and not nfree_all and col != 0 and cnstnd and \
            hasattr(be, "masked_grams_rgrad")
        nfree = be.size(x) if nfree_all else (-1 if late_count else be.count_free(iwhere))
        if nfree != 0 and col != 0:
            ok = True
            if nfree_all:
                free = None
            else:
                free = iwhere
            # r = -Z'(B(xcp - x) + g): its coefficients depend on the compact
            # matrices and the Cauchy point only, so a backend may form it in the
            # pass that computes the subspace matrix
            r = None
            fused = None
            if cnstnd or col == 0:
                mc = cm.bmv(c_vec)
                coef_y = mc[:col]
                coef_s = theta * mc[col:]
                if hasattr(be, "masked_grams_rgrad"):
                    kw = {}
                    if DEFER_REDUCED_GRADIENT and FUSE_SUBSPACE_STEP and USE_GRAM_RHS and \
                            getattr(be, "subspace_step_forms_r", False):
                        kw["want_r"] = False
                    if late_count:
                        fused = be.masked_grams_rgrad(ws, wy, free, z, x, g, theta,
                                                      coef_s, coef_y, count=True, **kw)
                        if fused is not None:
                            nfree = fused[-1]
                            fused = fused[:-1]
                    else:
                        fused = be.masked_grams_rgrad(ws, wy, free, z, x, g, theta,
                                                      coef_s, coef_y, **kw)
            if nfree < 0:
                nfree = be.count_free(iwhere)
        if nfree != 0 and col != 0:       # (a late count may have found none free)
            wtzr = None
            if fused is not None:
                yzzy, szzs, szzy, r, wtzr = fused
            else:
                yzzy, szzs, szzy = be.masked_grams(ws, wy, free)
            fac = form_k(cm, yzzy, szzs, szzy)
            if fac is None:
                ok = False
            rdef = None
            if ok and r is None:
                if not cnstnd and col > 0:
                    r = be.scale(g, -1.0)
                elif fused is not None and wtzr is not None:
                    # (deferred: the subspace step forms it; an array only if asked for)
                    rdef = (coef_y, coef_s,
                            lambda: be.reduced_gradient(z0, x, g, theta, ws, wy, coef_s,
                                                        coef_y, free))
                    z0 = z
                else:
                    r = be.reduced_gradient(z, x, g, theta, ws, wy, coef_s,
                                            coef_y, free)
            if ok:
                z, step = _subsm(be, z, r, x, g, lo, hi, ws, wy, cm, fac, free,
                                 wtzr if USE_GRAM_RHS else None, rdef)
            else:
                # refresh the memory and restart the iteration
                cm.reset()
                ws, wy = [], []
                updatd = False
                continue
        # ---------------- line search --------------------------------------
        if step is not None:
            d, dtd, gd, wtd_s, wtd_y, ratio = step   # (formed with the subspace step)
        else:
            d, dtd, gd = be.diff_dots(z, x, g)    # d = z - x, d'd, g'd
            wtd_s = wtd_y = ratio = None
        dnorm = math.sqrt(dtd)
        stpmx = BIG
        if cnstnd:
            if it == 0:
                stpmx = 1.0
            else:
                stpmx = min(BIG, ratio) if ratio is not None else \
                    be.max_step(x, d, lo, hi, BIG)
        if it == 0 and not boxed:
            stp = min(1.0 / dnorm, stpmx) if dnorm > 0 else stpmx
        else:
            stp = 1.0
        # (no backend operation writes into its arguments, so the old iterate
        # and gradient are kept by reference)
        xold = x
        gold = g
        ynew = rrnew = None
        fold = f
        gdold = gd
        restart = False
        if gd >= 0:
            # the search direction is not a descent direction
            restart = True
            ls_info = -4
        else:
            ls = LineSearch(f, gd, stp, 0.0, stpmx)
            ifun = 0
            iback = 0
            ls_info = 0
            while True:
                if ls.task[:5] == "ERROR":
                    ls_info = -3
                    restart = True
                    break
                if ls.task[:4] in ("CONV", "WARN"):
                    break
                # the search asks for f, g at the trial step
                ifun += 1
                iback = ifun - 1
                if iback >= maxls:
                    restart = True
                    break
                nfgv += 1
                stp = ls.stp
                if stp == 1.0:
                    x = z
                else:
                    x = be.lincomb2(stp, d, 1.0, xold)
                if ext is not None:
                    f, g, gd, pg, ynew, rrnew = ext(x, d, lo, hi, gold)
                else:
                    f, g = fun_and_grad(x)
                    gd = be.dot(g, d)
                    pg = ynew = rrnew = None
                ls.step(f, gd)
            stp = ls.stp
        if restart:
            x, g, f = xold, gold, fold
            if col == 0:
                info["task"] = "ABNORMAL_TERMINATION_IN_LNSRCH"
                break
            cm.reset()
            ws, wy = [], []
            updatd = False
            continue
        # ---------------- new iterate ---------------------------------------
        it += 1
        # (x, g are those of the search's last evaluation)
        sbgnrm = be.projgr(x, g, lo, hi) if pg is None else pg
        if it >= maxiter:
            info["task"] = "STOP: TOTAL NO. of ITERATIONS REACHED LIMIT"
            break
        if nfgv > maxfun:
            info["task"] = "STOP: TOTAL NO. of f AND g EVALUATIONS EXCEEDS LIMIT"
            break
        if sbgnrm <= pgtol:
            info["task"] = "CONVERGENCE: NORM_OF_PROJECTED_GRADIENT_<=_PGTOL"
            break
        ddum = max(abs(fold), abs(f), 1.0)
        if (fold - f) <= EPSMCH * factr * ddum:
            info["task"] = "CONVERGENCE: REL_REDUCTION_OF_F_<=_FACTR*EPSMCH"
            break
        # ---------------- BFGS update ---------------------------------------
        if ynew is not None:
            r, rr = ynew, rrnew                   # (came with the last evaluation)
        else:
            r, rr, _ = be.diff_dots(g, gold)      # y = g - g_old, y'y
        if stp == 1.0:
            dr = gd - gdold
            ddum = -gdold
        else:
            dr = (gd - gdold) * stp
            d = be.scale(d, stp)
            ddum = -gdold * stp
        if dr <= EPSMCH * ddum:
            nskip += 1
            updatd = False
            continue
        updatd = True
        if cm.col < m:
            cm.col += 1
        else:                                     # drop the oldest pair
            ws.pop(0)
            wy.pop(0)
            cm.ss[:m - 1, :m - 1] = cm.ss[1:, 1:]
            cm.sy[:m - 1, :m - 1] = cm.sy[1:, 1:]
            if wtd_s is not None:
                wtd_s, wtd_y = wtd_s[1:], wtd_y[1:]
        ws.append(d)
        wy.append(r)
        c = cm.col
        cm.theta = rr / dr
        if wtd_s is not None and len(wtd_s) == c - 1:
            # S_old^T d and d^T Y_old came with the subspace step (for the unit
            # step; the line search's step length scales both)
            sdots = wtd_s if stp == 1.0 else stp * wtd_s
            ydots = wtd_y if stp == 1.0 else stp * wtd_y
        else:
            both = be.dots(ws[:c - 1] + wy[:c - 1], d)   # one pass, one read-back
            sdots = both[:c - 1]                      # S_old^T d
            ydots = both[c - 1:]                      # d^T Y_old
        for j in range(c - 1):
            cm.sy[c - 1, j] = ydots[j]
            cm.ss[j, c - 1] = sdots[j]
        cm.ss[c - 1, c - 1] = dtd if stp == 1.0 else stp * stp * dtd
        cm.sy[c - 1, c - 1] = dr
        if not cm.form_t():
            cm.reset()
            ws, wy = [], []
            updatd = False
    info["nit"] = it
    info["nfev"] = nfgv
    info["fun"] = f
    info["jac"] = g               # (f, g belong to the point returned)
    info["pg"] = sbgnrm           # (of the same point: a failed search changes neither)
    return x, info


class _PathState(object):
    """Scalars and 2col-vectors carried along the projected-gradient path."""
    __slots__ = ("tj", "tsum", "nleft", "f1", "f2", "dtm", "p", "c", "t_done",
                 "i_done", "all_fixed")


def _cauchy(be, x, g, lo, hi, iwhere, ws, wy, cm, sbgnrm):
    """Generalized Cauchy point along the projected steepest-descent path.
    Returns (xcp, c = W^T (xcp - x), iwhere).

    The breakpoints the search crosses are walked with prefix sums (the
    recurrences for p, c, f', f'' are linear between stops), so millions of
    crossed bounds cost a few scans instead of one step each: on the device
    when the backend offers `breakpoint_walker` (sorted candidate table + hipCUB
    scans), else with NumPy cumsums over batches from `breakpoint_stream`.  The
    scalar walk is kept for the rare clamp of f'' and for the very last
    breakpoint, whose bookkeeping differs."""
    col, theta = cm.col, cm.theta
    if sbgnrm <= 0.0:
        return be.copy(x), np.zeros(2 * col), iwhere
    both = None
    if col > 0 and MERGE_READBACKS and hasattr(be, "cauchy_setup_dots"):
        # (the products W'd enqueued behind the set-up that forms d: one read-back)
        d, tbk, iwhere, st, both = be.cauchy_setup_dots(x, g, lo, hi, iwhere, wy + ws)
    else:
        d, tbk, iwhere, st = be.cauchy_setup(x, g, lo, hi, iwhere)
    nbreak = st["nbreak"]
    bnded = st["bnded"]
    STATS["cauchy_calls"] += 1
    STATS["breakpoints"] += nbreak
    S = _PathState()
    S.f1 = st["f1"]
    S.p = np.zeros(2 * col)
    if col > 0:
        if both is None:
            both = be.dots(wy + ws, d)             # one pass, one read-back
        both = np.asarray(both)
        S.p[:col] = both[:col]
        S.p[col:] = theta * both[col:]
    if not st["any_move"]:
        return be.copy(x), np.zeros(2 * col), iwhere
    S.c = np.zeros(2 * col)
    S.f2 = -theta * S.f1
    f2_org = S.f2
    mmat = None
    if col > 0:
        S.f2 -= float(np.dot(cm.bmv(S.p), S.p))
        mmat = middle_matrix(cm)                                   # v = mmat @ w
    S.dtm = -S.f1 / S.f2
    S.tsum = 0.0
    S.tj = 0.0
    S.t_done, S.i_done = -1.0, -1         # last breakpoint that was fixed
    S.all_fixed = False
    S.nleft = nbreak
    n_total = be.size(x)

    def scalar_walk(batch, k0):
        """One breakpoint at a time; returns True when the search is over."""
        bt, bi, bd, bx, bwy, bws = batch
        for k in range(k0, len(bt)):
            tj_new, ibp, dibp, xibp = bt[k], int(bi[k]), bd[k], bx[k]
            dt1 = tj_new - S.tj
            if S.dtm < dt1:
                return True
            S.tj = float(tj_new)
            S.tsum += dt1
            S.nleft -= 1
            S.t_done, S.i_done = float(tj_new), ibp
            zibp = (hi - xibp) if dibp > 0 else (lo - xibp)
            if S.nleft == 0 and nbreak == n_total:
                S.dtm = dt1
                S.all_fixed = True
                return True
            dibp2 = dibp * dibp
            S.f1 = S.f1 + dt1 * S.f2 + dibp2 - theta * dibp * zibp
            S.f2 = S.f2 - theta * dibp2
            if col > 0:
                S.c = S.c + dt1 * S.p
                wbp = np.concatenate((bwy[k], theta * bws[k]))
                v = mmat.dot(wbp)
                wmc = float(np.dot(S.c, v))
                wmp = float(np.dot(S.p, v))
                wmw = float(np.dot(wbp, v))
                S.p = S.p - dibp * wbp
                S.f1 += dibp * wmc
                S.f2 += 2.0 * dibp * wmp - dibp2 * wmw
            S.f2 = max(EPSMCH * f2_org, S.f2)
            if S.nleft > 0:
                S.dtm = -S.f1 / S.f2
            elif bnded:
                S.f1 = S.f2 = S.dtm = 0.0
                return True
            else:
                S.dtm = -S.f1 / S.f2
                return True
        return False

    def cumsum_walk(batch):
        """NumPy prefix-sum walk; returns (finished, first unprocessed k)."""
        bt, bi, bd, bx, bwy, bws = batch
        K = len(bt)
        kv = K if S.nleft > K else K - 1      # leave the globally last one
        if kv <= 0:
            return False, 0
        t = bt[:kv]
        dt = np.diff(np.concatenate(([S.tj], t)))
        dib = bd[:kv]
        dib2 = dib * dib
        zib = np.where(dib > 0, hi - bx[:kv], lo - bx[:kv])
        inc2 = -theta * dib2
        inc1x = dib2 - theta * dib * zib
        if col > 0:
            W = np.concatenate((bwy[:kv], theta * bws[:kv]), axis=1)
            P_after = S.p - np.cumsum(dib[:, None] * W, axis=0)
            P_before = np.vstack((S.p, P_after[:-1]))
            C_after = S.c + np.cumsum(dt[:, None] * P_before, axis=0)
            V = W.dot(mmat.T)
            inc2 = inc2 + 2.0 * dib * np.sum(P_before * V, axis=1) - \
                dib2 * np.sum(W * V, axis=1)
            inc1x = inc1x + dib * np.sum(C_after * V, axis=1)
        f2_after = S.f2 + np.cumsum(inc2)
        clamp = f2_after < EPSMCH * f2_org
        if np.any(clamp):
            kv = int(np.argmax(clamp))        # scalar walk from there
            f2_after = f2_after[:kv]
        if kv <= 0:
            return False, 0
        f2_before = np.concatenate(([S.f2], f2_after[:kv - 1]))
        f1_after = S.f1 + np.cumsum(dt[:kv] * f2_before + inc1x[:kv])
        dtm_after = -f1_after / f2_after
        dtm_before = np.concatenate(([S.dtm], dtm_after[:kv - 1]))
        stop = dtm_before < dt[:kv]
        stopped = bool(np.any(stop))
        kdone = int(np.argmax(stop)) if stopped else kv
        if kdone > 0:
            j = kdone - 1
            S.tj = S.tsum = float(t[j])
            S.nleft -= kdone
            S.t_done, S.i_done = float(t[j]), int(bi[j])
            S.f1, S.f2 = float(f1_after[j]), float(f2_after[j])
            S.dtm = float(dtm_after[j])
            if col > 0:
                S.p = P_after[j].copy()
                S.c = C_after[j].copy()
        return stopped, kdone

    if nbreak > 0:
        walker = None
        if hasattr(be, "breakpoint_walker"):
            walker = be.breakpoint_walker(tbk, d, ws, wy, theta, lo, hi,
                                          f2_org, mmat)
        fetch = None if walker else be.breakpoint_stream(tbk, d, ws, wy)
        finished = False
        while not finished:
            STATS["fetches"] += 1
            if walker:
                r = walker(S)             # advances S over a sorted window
                if r is None:
                    break                 # minimiser before the next breakpoint
                finished, tail = r
                if not finished and tail is not None:
                    finished = scalar_walk(tail, 0)
            else:
                batch = fetch(S.t_done, S.i_done, S.tsum + S.dtm)
                if batch is None:
                    break
                finished, k0 = cumsum_walk(batch)
                if not finished:
                    finished = scalar_walk(batch, k0)
        STATS["crossed"] += nbreak - S.nleft
    if not S.all_fixed:
        if S.dtm <= 0.0:
            S.dtm = 0.0
        S.tsum += S.dtm
    xcp, iwhere = be.cauchy_finish(x, d, tbk, lo, hi, iwhere, S.tsum,
                                   S.t_done, S.i_done, S.all_fixed)
    c = S.c
    if col > 0:
        c = c + S.dtm * S.p
    return xcp, c, iwhere


def _subsm(be, xcp, r, x, g, lo, hi, ws, wy, cm, fac, free, wtzr=None, rdef=None):
    """Subspace minimisation over the free variables at the Cauchy point with
    the projection refinement; returns the new point.  wtzr: [Y S]'Z r when the
    backend's Gram pass already produced it.  rdef (r is None then): (coef_y,
    coef_s, make_r) -- the W part of r for a backend whose subspace step forms r
    itself, and a callable that returns r as an array for the paths that need one."""
    col, theta = cm.col, cm.theta
    wv = np.zeros(2 * col)
    if wtzr is not None:
        both = np.asarray(wtzr)
    else:
        if r is None:
            r = rdef[2]()
        both = np.asarray(be.dots(wy + ws, r, free))   # one pass, one read-back
    wv[:col] = both[:col]
    wv[col:] = theta * both[col:]
    wv = solve_k(fac, wv, col)
    # d = (1/theta) r + (1/theta^2) Z'W wv   (wv already carries theta in S)
    # A backend may take the rest of the step -- the direction, its projection,
    # the line search's d = z - x with d'd and g'd, and the products of the stored
    # vectors with d that the BFGS update will ask for -- from ONE pass over them
    fused = None
    if FUSE_SUBSPACE_STEP and hasattr(be, "subspace_step"):
        if r is None:
            fused = be.subspace_step(None, ws, wy, wv[:col] / theta, wv[col:], theta,
                                     free, xcp, x, g, lo, hi, rdef=rdef[:2])
        else:
            fused = be.subspace_step(r, ws, wy, wv[:col] / theta, wv[col:], theta,
                                     free, xcp, x, g, lo, hi)
    if fused is not None:
        xnew, hit, dvec, dtd, gd, sd, yd, ratio = fused
        # (g'd of the projected point IS subsm's directional derivative dd_p)
        if not hit or gd <= 0.0:
            return xnew, (dvec, dtd, gd, sd, yd, ratio)
    if r is None:
        r = rdef[2]()
    d = be.subspace_direction(r, ws, wy, wv[:col] / theta, wv[col:], theta,
                              free)
    if fused is None:
        xnew, hit = be.project_step(xcp, d, lo, hi, free)
        if not hit:
            return xnew, None
        dd_p = be.dot_diff(xnew, x, g)
        if dd_p <= 0.0:
            return xnew, None
    # projected point is not a descent direction: truncate instead
    return be.truncated_step(xcp, d, lo, hi, free), None
