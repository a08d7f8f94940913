"""NumPy backend for nsol_amd.lbfgsb (TEST INFRASTRUCTURE): lets the CPU
suite run the L-BFGS-B iteration logic of the product against
scipy.optimize.minimize(method="L-BFGS-B") without a GPU.  The product uses
nsol_amd.lbfgsb_device.DeviceBackend, whose kernels are held to this backend in
the GPU tests."""
import numpy as np


class NumpyBackend(object):

    def size(self, x):
        return x.size

    def copy(self, x):
        return x.copy()

    def clip(self, x, lo, hi):
        return np.clip(np.asarray(x, dtype=np.float64), lo, hi)

    def scale(self, x, a):
        return a * x

    def lincomb2(self, a, x, b, y):
        return a * x + b * y

    def dot(self, x, y):
        return float(np.dot(x, y))

    def dots(self, vecs, v, free=None):
        if free is None:
            return [float(np.dot(w, v)) for w in vecs]
        msk = free <= 0
        return [float(np.dot(w[msk], v[msk])) for w in vecs]

    def dot_diff(self, a, b, g):
        return float(np.dot(a - b, g))

    def diff_dots(self, a, b, c=None):
        d = a - b
        return d, float(np.dot(d, d)), \
            (float(np.dot(d, c)) if c is not None else 0.0)

    def init_where(self, x, lo, hi):
        bounded = np.isfinite(lo) or np.isfinite(hi)
        if np.isfinite(lo) and np.isfinite(hi) and hi - lo <= 0:
            return np.full(x.size, 3, dtype=np.int8)
        return np.full(x.size, 0 if bounded else -1, dtype=np.int8)

    def count_free(self, iwhere):
        return int(np.count_nonzero(iwhere <= 0))

    def projgr(self, x, g, lo, hi):
        gi = g.copy()
        if np.isfinite(hi):
            neg = g < 0
            gi[neg] = np.maximum(x[neg] - hi, g[neg])
        if np.isfinite(lo):
            pos = g >= 0
            gi[pos] = np.minimum(x[pos] - lo, g[pos])
        return float(np.max(np.abs(gi))) if gi.size else 0.0

    def cauchy_setup(self, x, g, lo, hi, iwhere):
        has_lo, has_hi = np.isfinite(lo), np.isfinite(hi)
        neggi = -g
        iw = iwhere.copy()
        act = (iw != 3) & (iw != -1)
        tl = x - lo if has_lo else np.full(x.size, np.inf)
        tu = hi - x if has_hi else np.full(x.size, np.inf)
        xlower = has_lo & (tl <= 0)
        xupper = has_hi & (tu <= 0)
        new = np.zeros(x.size, dtype=np.int8)
        new[xlower & (neggi <= 0)] = 1
        sel = (~xlower) & xupper & (neggi >= 0)
        new[sel] = 2
        sel = (~xlower) & (~xupper) & (np.abs(neggi) <= 0)
        new[sel] = -3
        iw[act] = new[act]
        moving = (iw == 0) | (iw == -1)
        d = np.where(moving, neggi, 0.0)
        tbk = np.full(x.size, np.inf)
        bl = moving & has_lo & (neggi < 0)
        tbk[bl] = tl[bl] / (-neggi[bl])
        bu = moving & has_hi & (neggi > 0)
        tbk[bu] = tu[bu] / neggi[bu]
        nobk = moving & ~np.isfinite(tbk)
        st = {"f1": -float(np.dot(d, d)),
              "nbreak": int(np.count_nonzero(np.isfinite(tbk))),
              "bnded": not bool(np.any(nobk & (np.abs(neggi) > 0))),
              "any_move": bool(np.any(moving))}
        self._x = x
        return d, tbk, iw, st

    BATCH = 64

    def breakpoint_stream(self, tbk, d, ws, wy):
        order = np.lexsort((np.arange(tbk.size), tbk))
        order = order[np.isfinite(tbk[order])]
        pos = [0]
        x = self._x

        def fetch(t_done, i_done, t_hi):
            lim = t_hi * (1.0 + 1e-12) + 1e-300
            sel = []
            while pos[0] < order.size and len(sel) < self.BATCH:
                i = int(order[pos[0]])
                if tbk[i] > lim:
                    break
                sel.append(i)
                pos[0] += 1
            if not sel:
                return None
            sel = np.array(sel)
            return (tbk[sel].astype(np.float64), sel, d[sel], x[sel],
                    np.array([w[sel] for w in wy]).T.reshape(sel.size, len(wy)),
                    np.array([w[sel] for w in ws]).T.reshape(sel.size, len(ws)))
        return fetch

    def cauchy_finish(self, x, d, tbk, lo, hi, iwhere, tsum, t_done, i_done,
                      all_fixed):
        idx = np.arange(x.size)
        fixed = np.isfinite(tbk) & ((tbk < t_done) |
                                    ((tbk == t_done) & (idx <= i_done)))
        iw = iwhere.copy()
        xcp = x + tsum * np.where(fixed, 0.0, d)
        up = fixed & (d > 0)
        dn = fixed & (d <= 0)
        xcp[up] = hi
        xcp[dn] = lo
        iw[up] = 2
        iw[dn] = 1
        return xcp, iw

    def masked_grams(self, ws, wy, free):
        S = np.array(ws)
        Y = np.array(wy)
        if free is not None:
            msk = free <= 0
            S = S[:, msk]
            Y = Y[:, msk]
        return Y.dot(Y.T), S.dot(S.T), S.dot(Y.T)

    def reduced_gradient(self, z, x, g, theta, ws, wy, coef_s, coef_y, free):
        r = -theta * (z - x) - g
        for j in range(len(ws)):
            r = r + wy[j] * coef_y[j] + ws[j] * coef_s[j]
        if free is not None:
            r = np.where(free <= 0, r, 0.0)
        return r

    def subspace_direction(self, r, ws, wy, cy, cs, theta, free):
        d = r.copy()
        for j in range(len(ws)):
            d = d + wy[j] * cy[j] + ws[j] * cs[j]
        d = d * (1.0 / theta)
        if free is not None:
            d = np.where(free <= 0, d, 0.0)
        return d

    def project_step(self, xcp, d, lo, hi, free):
        msk = np.ones(xcp.size, bool) if free is None else (free <= 0)
        xn = xcp.copy()
        cand = np.clip(xcp + d, lo, hi)
        xn[msk] = cand[msk]
        hit = bool(np.any(msk & ((xn == lo) | (xn == hi))))
        return xn, hit

    def truncated_step(self, xcp, d, lo, hi, free):
        msk = np.ones(xcp.size, bool) if free is None else (free <= 0)
        ratio = np.full(xcp.size, np.inf)
        dn = msk & (d < 0) & np.isfinite(lo)
        up = msk & (d > 0) & np.isfinite(hi)
        t2 = lo - xcp
        ratio[dn] = np.where(t2[dn] >= 0, 0.0, t2[dn] / d[dn])
        t2 = hi - xcp
        ratio[up] = np.where(t2[up] <= 0, 0.0, t2[up] / d[up])
        alpha = 1.0
        ibd = -1
        if ratio.size:
            k = int(np.argmin(ratio))
            if ratio[k] < alpha:
                alpha, ibd = float(ratio[k]), k
        xn = xcp.copy()
        dd = d.copy()
        if alpha < 1.0 and ibd >= 0:
            xn[ibd] = hi if d[ibd] > 0 else lo
            dd[ibd] = 0.0
        xn[msk] = xn[msk] + alpha * dd[msk]
        return xn

    def max_step(self, x, d, lo, hi, big):
        stpmx = big
        if np.isfinite(lo):
            sel = d < 0
            a2 = lo - x[sel]
            if np.any(a2 >= 0):
                return 0.0
            if a2.size:
                stpmx = min(stpmx, float(np.min(a2 / d[sel])))
        if np.isfinite(hi):
            sel = d > 0
            a2 = hi - x[sel]
            if np.any(a2 <= 0):
                return 0.0
            if a2.size:
                stpmx = min(stpmx, float(np.min(a2 / d[sel])))
        return stpmx
