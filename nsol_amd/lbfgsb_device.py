"""Device backend of nsol_amd.lbfgsb: every length-n operation of the
L-BFGS-B iteration is a HIP kernel of libnsol_hip.so (nsol_lbfgsb.hip) on
torch HIP tensors; only scalars, <= 2m x 2m matrices and the (few) breakpoints
the Cauchy search actually crosses travel to the host."""
import ctypes

import numpy as np
import torch

from . import _lib, _timing, ops
from .device import suffix, stream_ptr


# The Cauchy search's window of breakpoints reaches to the minimiser of the current
# segment; crossing bounds moves that minimiser outwards, so a window stretched by
# this factor more often holds the end of the search and saves the next window's
# compaction sweep over all n variables (at the price of a longer sort)
WINDOW_STRETCH = 1.5       # (config 4's Huber run: 52 -> 30 windows, same iterates)


def _fn(name, t):
    fn = getattr(_lib.load(), "nsol_lb_%s_%s" % (name, suffix(t)))
    kt = _timing.active()              # (measurement only: bench.py / bench_admm.py)
    return fn if kt is None else kt.wrap("lb_" + name, fn)


def _p(t):
    return None if t is None else t.data_ptr()


class DeviceBackend(object):

    # Breakpoints per window of the Cauchy search.  A window costs one compaction
    # sweep over all n variables however few it keeps, so large windows pay: at
    # 512^3 (about 2 M breakpoints crossed per search) the Huber run of config 4
    # takes 0.388 / 0.361 / 0.337 / 0.330 s with 0.5 / 1 / 4 / 16 Mi per window.
    # The walk's tables need (8 + 8 col) doubles per breakpoint: the window is
    # also held to TABLE_BYTES of them.
    CAPACITY = 1 << 24
    TABLE_BYTES = 4 << 30

    def __init__(self):
        self._res = None
        self._tmp = None
        self._gram_ws = None

    # ---- small helpers
    def _bufs(self, like):
        ws, _ = ops._workspace(like.device)
        if self._res is None or self._res.device != like.device:
            self._res = torch.empty(8, dtype=torch.float64, device=like.device)
        return ws, self._res

    def _check(self, rc, what):
        _lib.check(rc, "nsol_lb_" + what)

    # lbfgsb.minimize pads the problem to a multiple of this many variables (the
    # kernels' 16-byte vectors and 16 mask bytes per access; an odd-sized volume
    # would otherwise run every pass element by element)
    pad_to = 16

    def pad(self, x, n_pad):
        out = torch.zeros(n_pad, dtype=x.dtype, device=x.device)
        out[:x.numel()].copy_(x)
        return out

    def head(self, x, n):
        return x[:n]

    def size(self, x):
        return x.numel()

    def copy(self, x):
        return x.clone()

    def clip(self, x, lo, hi):
        return ops.clip(x, lo, hi)

    def scale(self, x, a):
        return ops.scale(x, a)

    def lincomb2(self, a, x, b, y):
        return ops.lincomb2(a, x, b, y)

    def dot(self, x, y):
        return ops.dot(x, y)

    def dot_diff(self, a, b, g):
        return self.diff_dots(a, b, g)[2]

    def diff_dots(self, a, b, c=None):
        """(a - b, its squared norm, its product with c) from one pass."""
        ws, res = self._bufs(a)
        out = torch.empty_like(a)
        self._check(_fn("diff_dots", a)(
            _p(a), _p(b), _p(c) if c is not None else None, _p(out), a.numel(),
            _p(res), _p(ws), stream_ptr()), "diff_dots")
        r = res[:2].cpu().numpy()
        return out, float(r[0]), float(r[1])

    def init_where(self, x, lo, hi):
        bounded = np.isfinite(lo) or np.isfinite(hi)
        val = 0 if bounded else -1
        if np.isfinite(lo) and np.isfinite(hi) and hi - lo <= 0:
            val = 3
        return torch.full((x.numel(),), val, dtype=torch.int8,
                          device=x.device)

    def count_free(self, iwhere):
        ws, res = self._bufs(iwhere)
        self._check(_lib.load().nsol_lb_count_free(
            _p(iwhere), iwhere.numel(), _p(res), _p(ws), stream_ptr()),
            "count_free")
        return int(round(float(res[0].item())))

    def projgr(self, x, g, lo, hi):
        ws, res = self._bufs(x)
        self._check(_fn("projgr", x)(_p(x), _p(g), x.numel(), float(lo),
                                     float(hi), _p(res), _p(ws),
                                     stream_ptr()), "projgr")
        return float(res[0].item())

    def dots(self, vecs, v, free=None):
        """[sum_free vecs[k]*v]: one pass over v and the vectors, one read-back."""
        import ctypes
        if not vecs:
            return []
        lib = _lib.load()
        if self._gram_ws is None or self._gram_ws.device != v.device:
            self._gram_ws = torch.empty(int(lib.nsol_lb_gram_ws_doubles()),
                                        dtype=torch.float64, device=v.device)
        out = torch.empty(len(vecs), dtype=torch.float64, device=v.device)
        ptrs = (ctypes.c_void_p * len(vecs))(*[w.data_ptr() for w in vecs])
        self._check(_fn("mdots", v)(ptrs, len(vecs), _p(v), _p(free), v.numel(),
                                    _p(out), _p(self._gram_ws), stream_ptr()),
                    "mdots")
        return [float(t) for t in out.cpu().numpy()]

    USE_GRAM_KERNEL = True    # False: one masked dot per matrix entry

    def masked_grams(self, ws_list, wy_list, free):
        """Y'ZZ'Y, S'ZZ'S, S'ZZ'Y over the free variables (Z), col x col."""
        c = len(ws_list)
        like = ws_list[0]
        if self.USE_GRAM_KERNEL and 2 * c <= 24:
            return self._masked_grams_one_pass(ws_list, wy_list, free)
        wsb, _ = self._bufs(like)
        out = torch.empty(3 * c * c, dtype=torch.float64, device=like.device)
        fn = _fn("mdot", like)
        n = like.numel()

        def launch(a, b, slot):
            self._check(fn(_p(a), _p(b), _p(free), n,
                           out.data_ptr() + 8 * slot, _p(wsb), stream_ptr()),
                        "mdot")
        for i in range(c):
            for j in range(i, c):
                launch(wy_list[i], wy_list[j], i * c + j)
                launch(ws_list[i], ws_list[j], c * c + i * c + j)
            for j in range(c):
                launch(ws_list[i], wy_list[j], 2 * c * c + i * c + j)
        h = out.cpu().numpy().reshape(3, c, c)
        yy = np.triu(h[0]) + np.triu(h[0], 1).T
        ss = np.triu(h[1]) + np.triu(h[1], 1).T
        return yy, ss, h[2]

    # subspace_step can form r = Z(-theta (xcp - x) - g + W M c) itself (rdef=...): the
    # Gram pass then runs with want_r=False and r never exists as an array
    subspace_step_forms_r = True

    def masked_grams_rgrad(self, ws_list, wy_list, free, z, x, g, theta, coef_s,
                           coef_y, count=False, want_r=True):
        """masked_grams and reduced_gradient from ONE pass over the 2c vectors
        (nsol_lb_masked_gram_rgrad_*), with [Y S]'Z r (Y first) for the subspace
        right-hand side; None where that kernel does not apply.  count: the number
        of free variables (count_free) enqueued ahead and read back with the rest,
        appended to the result."""
        import ctypes
        c = len(ws_list)
        vecs = list(wy_list) + list(ws_list)           # Y first, then S
        like = vecs[0]
        nv = 2 * c
        if not self.USE_GRAM_KERNEL or nv > 24 or free is None:
            return None
        lib = _lib.load()
        if self._gram_ws is None or self._gram_ws.device != like.device:
            self._gram_ws = torch.empty(int(lib.nsol_lb_gram_ws_doubles()),
                                        dtype=torch.float64, device=like.device)
        npairs = nv * (nv + 1) // 2
        out = torch.empty(npairs + nv + 1, dtype=torch.float64, device=like.device)
        r = torch.empty_like(like) if want_r else None
        ptrs = (ctypes.c_void_p * nv)(*[v.data_ptr() for v in vecs])
        base = (ctypes.c_void_p * 3)(z.data_ptr(), x.data_ptr(), g.data_ptr())
        bco = np.array([-theta, theta, -1.0], dtype=np.float64)
        wco = np.ascontiguousarray(list(coef_y) + list(coef_s), dtype=np.float64)
        if count:
            wsb, _ = self._bufs(like)
            self._check(lib.nsol_lb_count_free(
                _p(free), free.numel(), out.data_ptr() + 8 * (npairs + nv), _p(wsb),
                stream_ptr()), "count_free")
        rc = _fn("masked_gram_rgrad", like)(
            ptrs, nv, _p(free), like.numel(), _p(out), _p(self._gram_ws),
            ctypes.cast(base, ctypes.c_void_p), bco.ctypes.data, wco.ctypes.data,
            _p(r), stream_ptr())
        if rc == -2:
            return None
        self._check(rc, "masked_gram_rgrad")
        flat = out.cpu().numpy()
        gm = np.zeros((nv, nv))
        gm[np.triu_indices(nv)] = flat[:npairs]
        gm = gm + np.triu(gm, 1).T
        # W'Z r for r = Z (b + W wco): from the matrix and the products with b
        wtzr = flat[npairs:npairs + nv] + gm @ wco
        res = (gm[:c, :c], gm[c:, c:], gm[c:, :c], r, wtzr)
        return res + (int(round(float(flat[npairs + nv]))),) if count else res

    def _masked_grams_one_pass(self, ws_list, wy_list, free):
        """All entries from ONE pass over the 2c vectors (nsol_lb_masked_gram_*)."""
        import ctypes
        c = len(ws_list)
        vecs = list(wy_list) + list(ws_list)           # Y first, then S
        like = vecs[0]
        nv = 2 * c
        lib = _lib.load()
        if self._gram_ws is None or self._gram_ws.device != like.device:
            self._gram_ws = torch.empty(int(lib.nsol_lb_gram_ws_doubles()),
                                        dtype=torch.float64, device=like.device)
        npairs = nv * (nv + 1) // 2
        out = torch.empty(npairs, dtype=torch.float64, device=like.device)
        ptrs = (ctypes.c_void_p * nv)(*[v.data_ptr() for v in vecs])
        self._check(_fn("masked_gram", like)(
            ptrs, nv, _p(free), like.numel(), _p(out), _p(self._gram_ws),
            stream_ptr()), "masked_gram")
        flat = out.cpu().numpy()
        g = np.zeros((nv, nv))
        g[np.triu_indices(nv)] = flat                  # row-major pairs, i <= j
        g = g + np.triu(g, 1).T
        return g[:c, :c], g[c:, c:], g[c:, :c]         # Y'Y, S'S, S'Y

    # ---- Cauchy point
    def cauchy_setup(self, x, g, lo, hi, iwhere):
        ws, res = self._bufs(x)
        d = torch.empty_like(x)
        tbk = torch.empty_like(x)
        iw = iwhere.clone()
        self._check(_fn("cauchy_setup", x)(
            _p(x), _p(g), x.numel(), float(lo), float(hi), _p(iw), _p(d),
            _p(tbk), _p(res), _p(ws), stream_ptr()), "cauchy_setup")
        r = res[:4].cpu().numpy()
        st = {"f1": -float(r[0]), "nbreak": int(round(r[1])),
              "bnded": int(round(r[2])) == 0, "any_move": int(round(r[3])) > 0}
        self._x = x
        return d, tbk, iw, st

    def cauchy_setup_dots(self, x, g, lo, hi, iwhere, vecs):
        """cauchy_setup and dots(vecs, d) enqueued one behind the other, their
        scalars read back together."""
        lib = _lib.load()
        ws, _ = self._bufs(x)
        nv = len(vecs)
        both = torch.empty(4 + nv, dtype=torch.float64, device=x.device)
        d = torch.empty_like(x)
        tbk = torch.empty_like(x)
        iw = iwhere.clone()
        self._check(_fn("cauchy_setup", x)(
            _p(x), _p(g), x.numel(), float(lo), float(hi), _p(iw), _p(d),
            _p(tbk), _p(both), _p(ws), stream_ptr()), "cauchy_setup")
        if self._gram_ws is None or self._gram_ws.device != x.device:
            self._gram_ws = torch.empty(int(lib.nsol_lb_gram_ws_doubles()),
                                        dtype=torch.float64, device=x.device)
        ptrs = (ctypes.c_void_p * nv)(*[w.data_ptr() for w in vecs])
        self._check(_fn("mdots", d)(ptrs, nv, _p(d), None, d.numel(),
                                    both.data_ptr() + 32, _p(self._gram_ws),
                                    stream_ptr()), "mdots")
        r = both.cpu().numpy()
        st = {"f1": -float(r[0]), "nbreak": int(round(r[1])),
              "bnded": int(round(r[2])) == 0, "any_move": int(round(r[3])) > 0}
        self._x = x
        return d, tbk, iw, st, [float(t) for t in r[4:]]

    def breakpoint_walker(self, tbk, d, ws_list, wy_list, theta, lo, hi,
                          f2_org, mmat):
        """Device walk of the crossed breakpoints: window -> compaction ->
        radix sort by (t, index) -> prefix-sum walk (nsol_lb_cauchy_walk_*).
        The returned callable advances the path state in place and answers
        None (no breakpoint before the minimiser), (True, None) (search over) or
        (False, tail) where `tail` is a one-element host batch that needs the
        scalar rule (clamped f'' or the globally last breakpoint)."""
        x = self._x
        dev = tbk.device
        n = tbk.numel()
        col = len(ws_list)
        cap = min(self.CAPACITY, n,
                  max(1 << 20, self.TABLE_BYTES // ((8 + 8 * col) * 8)))
        f32 = tbk.dtype == torch.float32
        lib = _lib.load()
        idx = torch.empty(cap, dtype=torch.int64, device=dev)
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        event = torch.zeros(2, dtype=torch.int32, device=dev)
        out = torch.empty(8 + 4 * max(col, 1), dtype=torch.float64, device=dev)
        sel = _fn("select", tbk)
        gat = _fn("gather", tbk)
        srt = _fn("sort_candidates", tbk)
        walk = _fn("cauchy_walk", tbk)
        PW = ctypes.c_void_p * max(col, 1)
        wy_p = PW(*[w.data_ptr() for w in wy_list])
        ws_p = PW(*[w.data_ptr() for w in ws_list])
        buf = {}

        def scratch(name, count, dtype):
            t = buf.get(name)
            if t is None or t.numel() < count:
                t = torch.empty(int(count), dtype=dtype, device=dev)
                buf[name] = t
            return t

        def host_rows(k_from, k_to):
            sub = idx[k_from:k_to].contiguous()
            m = k_to - k_from

            def g(src):
                o = torch.empty(m, dtype=src.dtype, device=dev)
                self._check(gat(_p(src), _p(sub), m, _p(o), stream_ptr()),
                            "gather")
                return o.cpu().numpy().astype(np.float64)
            wyv = np.stack([g(w) for w in wy_list], 1) if col else \
                np.zeros((m, 0))
            wsv = np.stack([g(w) for w in ws_list], 1) if col else \
                np.zeros((m, 0))
            return (g(tbk), sub.cpu().numpy(), g(d), g(x), wyv, wsv)

        def advance(S):
            lim = (S.tsum + S.dtm) * (1.0 + (1e-6 if f32 else 1e-12)) + 1e-300
            if WINDOW_STRETCH != 1.0:
                lim = S.tsum + WINDOW_STRETCH * (lim - S.tsum)
            # the compaction counts as it goes (its counter runs past the
            # capacity): a window that overflows is halved and selected again
            while True:
                self._check(sel(_p(tbk), n, float(S.t_done), int(S.i_done),
                                float(lim), _p(idx), cap, _p(cnt),
                                stream_ptr()), "select")
                count = int(cnt.item())
                if count <= cap:
                    break
                # shrink in proportion to the overflow (breakpoints are dense
                # and fairly even on image data), at least by half
                base = max(S.t_done, 0.0)
                lim = base + min(0.5, 0.8 * cap / count) * (lim - base)
            if count == 0:
                return None
            tmp = scratch("sort", lib.nsol_lb_sort_tmp_bytes(
                count, 4 if f32 else 8), torch.uint8)
            self._check(srt(_p(tbk), _p(idx), count, _p(tmp), tmp.numel(),
                            stream_ptr()), "sort_candidates")
            kv = count if S.nleft > count else count - 1
            kdone, stopped, clamp_at = 0, False, -1
            if kv > 0:
                table = scratch("table", lib.nsol_lb_walk_table_doubles(
                    kv, col), torch.float64)
                stmp = scratch("scan", lib.nsol_lb_walk_tmp_bytes(kv),
                               torch.uint8)
                params = torch.from_numpy(np.concatenate(
                    (S.p, S.c, mmat.reshape(-1) if col else np.zeros(0)))
                ).to(dev) if col else None
                self._check(walk(
                    _p(tbk), _p(d), _p(x), _p(idx), kv,
                    ctypes.cast(wy_p, ctypes.c_void_p),
                    ctypes.cast(ws_p, ctypes.c_void_p), col, float(theta),
                    float(lo), float(hi), float(S.tj), float(S.f1),
                    float(S.f2), float(f2_org), float(S.dtm), _p(params),
                    _p(table), _p(stmp), stmp.numel(), _p(event), _p(out),
                    stream_ptr()), "cauchy_walk")
                o = out.cpu().numpy()
                kdone, stopped, clamp_at = int(o[0]), o[1] > 0.5, int(o[2])
                if kdone > 0:
                    S.tj = S.tsum = float(o[3])
                    S.nleft -= kdone
                    S.t_done, S.i_done = float(o[3]), int(o[4])
                    S.f1, S.f2, S.dtm = float(o[5]), float(o[6]), float(o[7])
                    if col:
                        S.p = o[8:8 + 2 * col].copy()
                        S.c = o[8 + 2 * col:8 + 4 * col].copy()
            if stopped:
                return True, None
            if kdone < count and (kdone == clamp_at or kdone == kv):
                # clamped f'' or the globally last breakpoint: scalar rule
                return False, host_rows(kdone, kdone + 1)
            return False, None
        return advance

    def cauchy_finish(self, x, d, tbk, lo, hi, iwhere, tsum, t_done, i_done,
                      all_fixed):
        xcp = torch.empty_like(x)
        self._check(_fn("cauchy_finish", x)(
            _p(x), _p(d), _p(tbk), x.numel(), float(lo), float(hi),
            _p(iwhere), _p(xcp), float(tsum), float(t_done), int(i_done),
            stream_ptr()), "cauchy_finish")
        return xcp, iwhere

    # ---- limited-memory combinations
    def _wcomb(self, like, free, scale, base, bcoef, wvecs, wcoef):
        out = torch.empty_like(like)
        PT = ctypes.c_void_p * max(len(base), 1)
        PW = ctypes.c_void_p * max(len(wvecs), 1)
        bp = PT(*[b.data_ptr() for b in base])
        wp = PW(*[w.data_ptr() for w in wvecs])
        bc = np.ascontiguousarray(bcoef, dtype=np.float64)
        wc = np.ascontiguousarray(wcoef if len(wcoef) else [0.0],
                                  dtype=np.float64)
        self._check(_fn("wcomb", like)(
            _p(out), like.numel(), _p(free), float(scale), len(base),
            ctypes.cast(bp, ctypes.c_void_p), bc.ctypes.data, len(wvecs),
            ctypes.cast(wp, ctypes.c_void_p), wc.ctypes.data, stream_ptr()),
            "wcomb")
        return out

    def reduced_gradient(self, z, x, g, theta, ws_list, wy_list, coef_s,
                         coef_y, free):
        return self._wcomb(x, free, 1.0, [z, x, g], [-theta, theta, -1.0],
                           list(wy_list) + list(ws_list),
                           list(coef_y) + list(coef_s))

    def subspace_direction(self, r, ws_list, wy_list, cy, cs, theta, free):
        return self._wcomb(r, free, 1.0 / theta, [r], [1.0],
                           list(wy_list) + list(ws_list),
                           list(cy) + list(cs))

    def subspace_step(self, r, ws_list, wy_list, cy, cs, theta, free, xcp, x, g,
                      lo, hi, rdef=None):
        """subspace_direction + project_step + diff_dots(xn, x, g) + the products
        of every stored vector with d = xn - x, from one pass over the stored
        vectors.  Returns (xn, hit, d, d'd, g'd, S'd, Y'd, ratio) -- ratio: the
        smallest feasible step ratio along d from x (inf when nothing limits the
        step) -- or None when the fused kernel does not apply (nothing launched)."""
        vecs = list(wy_list) + list(ws_list)
        nw = len(vecs)
        if nw < 1:
            return None
        lib = _lib.load()
        if self._gram_ws is None or self._gram_ws.device != xcp.device:
            self._gram_ws = torch.empty(int(lib.nsol_lb_gram_ws_doubles()),
                                        dtype=torch.float64, device=xcp.device)
        PW = ctypes.c_void_p * nw
        wp = PW(*[w.data_ptr() for w in vecs])
        wc = np.ascontiguousarray(list(cy) + list(cs), dtype=np.float64)
        out = torch.empty(nw + 4, dtype=torch.float64, device=xcp.device)
        xn = torch.empty_like(xcp)
        d = torch.empty_like(xcp)
        if r is None:
            # rdef = (coef_y, coef_s) of r's W part; its base part is scipy cmprlb's
            # -theta (xcp - x) - g  (what reduced_gradient / masked_grams_rgrad use)
            rb = np.array([-theta, theta, -1.0], dtype=np.float64)
            rw = np.ascontiguousarray(list(rdef[0]) + list(rdef[1]), dtype=np.float64)
            rc = _fn("subspace_step_r", xcp)(
                ctypes.cast(wp, ctypes.c_void_p), wc.ctypes.data, nw, rb.ctypes.data,
                rw.ctypes.data, _p(xcp), _p(x), _p(g), _p(free), xcp.numel(),
                1.0 / theta, float(lo), float(hi), _p(xn), _p(d), _p(out),
                _p(self._gram_ws), stream_ptr())
        else:
            rc = _fn("subspace_step", r)(
                ctypes.cast(wp, ctypes.c_void_p), wc.ctypes.data, nw, _p(r), _p(xcp),
                _p(x), _p(g), _p(free), r.numel(), 1.0 / theta, float(lo), float(hi),
                _p(xn), _p(d), _p(out), _p(self._gram_ws), stream_ptr())
        if rc == -2:
            return None
        self._check(rc, "subspace_step")
        res = out.cpu().numpy()
        c = len(ws_list)
        return (xn, res[0] > 0, d, float(res[1]), float(res[2]),
                res[3 + c:3 + 2 * c].copy(), res[3:3 + c].copy(),
                -float(res[3 + 2 * c]))

    # ---- subspace step and line-search bound
    def project_step(self, xcp, d, lo, hi, free):
        ws, res = self._bufs(xcp)
        xn = torch.empty_like(xcp)
        self._check(_fn("project_step", xcp)(
            _p(xcp), _p(d), xcp.numel(), float(lo), float(hi), _p(free),
            _p(xn), _p(res), _p(ws), stream_ptr()), "project_step")
        return xn, float(res[0].item()) > 0

    def _ratio_min(self, x, d, lo, hi, free):
        ws, res = self._bufs(x)
        self._check(_fn("ratio_min", x)(
            _p(x), _p(d), x.numel(), float(lo), float(hi), _p(free),
            _p(res), _p(ws), stream_ptr()), "ratio_min")
        r = res[:2].cpu().numpy()
        return float(r[0]), int(r[1])

    def truncated_step(self, xcp, d, lo, hi, free):
        ratio, k = self._ratio_min(xcp, d, lo, hi, free)
        alpha, ibd = 1.0, -1
        if k >= 0 and ratio < alpha:
            alpha, ibd = ratio, k
        xn = torch.empty_like(xcp)
        self._check(_fn("trunc_apply", xcp)(
            _p(xcp), _p(d), xcp.numel(), float(lo), float(hi), _p(free),
            float(alpha), int(ibd if alpha < 1.0 else -1), _p(xn),
            stream_ptr()), "trunc_apply")
        return xn

    def max_step(self, x, d, lo, hi, big):
        ratio, k = self._ratio_min(x, d, lo, hi, None)
        return min(big, ratio) if k >= 0 else big
