// Stand-alone operators of the NSoL hot path for gfx950: finite differences,
// element-wise proxes / axpys, deterministic reductions, the ADMM outer update
// and the robust data term.  All are HBM-bound maps: one coalesced pass, x is
// the fastest-varying thread index, neighbours along y/z come from other rows
// that the same or an adjacent wave touches (L1/L2 hits).
#include <type_traits>

#include "nsol_common.hpp"
#include "nsol_stencil.hpp"

using namespace nsol;

namespace {

// ---------------------------------------------------------------- grad ----
// reference: linear_operators.py:98-106 (D_a = convolve(x, [1,-1]/h, "constant"))
template <typename T, int VEC, int ROWS, bool RAG>
__global__ __launch_bounds__(kBlock) void k_grad(const T *__restrict__ x,
                                                  T *__restrict__ g, Geom<T> G) {
  const int64_t nrg = row_groups<T, VEC, ROWS>(G);
  for (int64_t rg = blockIdx.y; rg < nrg; rg += gridDim.y) {
  const Voxel c = voxel_at<T, VEC, ROWS>(G, rg);
  if (!c.ok) continue;
  T v[VEC], hi[VEC], d[VEC];
  vlc<RAG, T, VEC>(c, x + c.i, v);
  const T right = (c.ix + VEC < G.nx) ? x[c.i + VEC] : T(0);
  fwd_diff_x<T, VEC>(v, right, G.wx, d);
  vs<RAG, T, VEC>(c.nval, g + c.i, d);
  if (G.ndim >= 2) {
    vzero(hi);
    if (c.iy + 1 < G.ny) vlc<RAG, T, VEC>(c, x + c.i + G.sy, hi);
    fwd_diff<T, VEC>(v, hi, G.wy, d);
    vs<RAG, T, VEC>(c.nval, g + G.n + c.i, d);
  }
  if (G.ndim >= 3) {
    vzero(hi);
    if (c.iz + 1 < G.nz) vlc<RAG, T, VEC>(c, x + c.i + G.sz, hi);
    fwd_diff<T, VEC>(v, hi, G.wz, d);
    vs<RAG, T, VEC>(c.nval, g + 2 * G.n + c.i, d);
  }
  }
}

// K^T at the lane's VEC voxels: sum over a of p_a[i]*(-w_a) + p_a[i-e_a]*w_a,
// accumulated x, y, z as linear_operators.py:158-169 does (`D_adj_x += ...`)
template <bool RAG, typename T, int VEC>
__device__ __forceinline__ void grad_adj_vec(const T *__restrict__ p,
                                             const Geom<T> &G, const Voxel &c,
                                             T (&acc)[VEC]) {
  T v[VEC], lo[VEC];
  vlc<RAG, T, VEC>(c, p + c.i, v);
  const T left = (c.ix > 0) ? p[c.i - 1] : T(0);
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    const T l = (k > 0) ? v[(k + VEC - 1) % VEC] : left;
    acc[k] = v[k] * (-G.wx) + l * G.wx;
  }
  if (G.ndim >= 2) {
    const T *py = p + G.n;
    vlc<RAG, T, VEC>(c, py + c.i, v);
    vzero(lo);
    if (c.iy > 0) vlc<RAG, T, VEC>(c, py + c.i - G.sy, lo);
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] += v[k] * (-G.wy) + lo[k] * G.wy;
  }
  if (G.ndim >= 3) {
    const T *pz = p + 2 * G.n;
    vlc<RAG, T, VEC>(c, pz + c.i, v);
    vzero(lo);
    if (c.iz > 0) vlc<RAG, T, VEC>(c, pz + c.i - G.sz, lo);
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] += v[k] * (-G.wz) + lo[k] * G.wz;
  }
}

// AXPY: out = x - tau * K^T p, the argument of prox_f in one pass
// (primal_dual_solver.py:246-248) when prox_f itself is not one of the fused ones
template <typename T, int VEC, int ROWS, bool RAG, bool AXPY = false>
__global__ __launch_bounds__(kBlock) void k_grad_adj(const T *__restrict__ p,
                                                      T *__restrict__ out,
                                                      Geom<T> G,
                                                      const T *__restrict__ x = nullptr,
                                                      T tau = T(0)) {
  const int64_t nrg = row_groups<T, VEC, ROWS>(G);
  for (int64_t rg = blockIdx.y; rg < nrg; rg += gridDim.y) {
    const Voxel c = voxel_at<T, VEC, ROWS>(G, rg);
    if (!c.ok) continue;
    T acc[VEC];
    grad_adj_vec<RAG, T, VEC>(p, G, c, acc);
    if constexpr (AXPY) {
      T xv[VEC];
      vlc<RAG, T, VEC>(c, x + c.i, xv);
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] = xv[k] - tau * acc[k];
    }
    vs<RAG, T, VEC>(c.nval, out + c.i, acc);
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_diff_axis(const T *__restrict__ x,
                                                       T *__restrict__ out,
                                                       Geom<T> G, int dir,
                                                       int adjoint, T w) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t step = dir == 0 ? 1 : (dir == 1 ? G.sy : G.sz);
  const int64_t len = dir == 0 ? G.nx : (dir == 1 ? G.ny : G.nz);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < G.n;
       i += stride) {
    const int64_t ix = i % G.nx;
    const int64_t r = i / G.nx;
    const int64_t pos = dir == 0 ? ix : (dir == 1 ? r % G.ny : r / G.ny);
    const T c = x[i];
    if (!adjoint) {
      const T nb = (pos + 1 < len) ? x[i + step] : T(0);
      out[i] = nb * w + c * (-w);
    } else {
      const T nb = (pos > 0) ? x[i - step] : T(0);
      out[i] = c * (-w) + nb * w;
    }
  }
}

// --------------------------------------------------------- element-wise ----
template <typename T, typename F>
__global__ __launch_bounds__(kBlock) void k_map(int64_t n, F f) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += stride)
    f(i);
}

template <typename T, typename F>
inline int launch_map(int64_t n, void *stream, F f) {
  if (n < 0) return NSOL_EINVAL;
  if (n == 0) return 0;
  hipLaunchKernelGGL((k_map<T, F>), dim3(grid_for(n)), dim3(kBlock), 0,
                     as_stream(stream), n, f);
  return launch_status();
}

// ------------------------------------------------------------ reductions ----
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  return v;
}

// block partial -> ws[blockIdx]; deterministic (fixed tree, fixed order)
__device__ __forceinline__ void block_store_partial(double v, double *ws) {
  __shared__ double s[kBlock / kWave];
  v = wave_sum(v);
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  if (lane == 0) s[wv] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < kBlock / kWave; ++k) t += s[k];
    ws[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(kBlock) void k_reduce_final(const double *ws,
                                                          int nparts,
                                                          double *result,
                                                          double scale) {
  double v = 0.0;
  for (int k = threadIdx.x; k < nparts; k += kBlock) v += ws[k];
  __shared__ double s[kBlock / kWave];
  v = wave_sum(v);
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  if (lane == 0) s[wv] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < kBlock / kWave; ++k) t += s[k];
    result[0] = t * scale;
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_dot(const T *__restrict__ x,
                                                 const T *__restrict__ y,
                                                 int64_t n, double *ws) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += stride)
    acc += (double)x[i] * (double)y[i];
  block_store_partial(acc, ws);
}

inline int reduce_grid(int64_t n) {
  int g = grid_for(n);
  return g > kReducePartials ? kReducePartials : g;
}

// ------------------------------------------------------------------ ADMM ----
// admm_linear_solver.py:208-216 with grad fused in
// NORM: also the sum of squares of the right-hand side written (per workgroup:
// ws[block]) -- the lower part of ||b|| that LSMR starts from
template <typename T, int VEC, int ROWS, bool RAG, bool NORM = false>
__global__ __launch_bounds__(kBlock) void k_admm_vw(const T *__restrict__ x,
                                                     T *__restrict__ v,
                                                     T *__restrict__ w,
                                                     const T *__restrict__ c,
                                                     T *__restrict__ rhs,
                                                     Geom<T> G, T thr,
                                                     T rhs_scale, double *ws = nullptr) {
  double acc = 0.0;
  const int64_t nrg = row_groups<T, VEC, ROWS>(G);
  for (int64_t rg = blockIdx.y; rg < nrg; rg += gridDim.y) {
  const Voxel q = voxel_at<T, VEC, ROWS>(G, rg);
  if (!q.ok) continue;
  T xc[VEC], hi[VEC];
  T t[3][VEC], cc[3][VEC];
  vlc<RAG, T, VEC>(q, x + q.i, xc);
  const T right = (q.ix + VEC < G.nx) ? x[q.i + VEC] : T(0);
  fwd_diff_x<T, VEC>(xc, right, G.wx, t[0]);
  if (G.ndim >= 2) {
    vzero(hi);
    if (q.iy + 1 < G.ny) vlc<RAG, T, VEC>(q, x + q.i + G.sy, hi);
    fwd_diff<T, VEC>(xc, hi, G.wy, t[1]);
  }
  if (G.ndim >= 3) {
    vzero(hi);
    if (q.iz + 1 < G.nz) vlc<RAG, T, VEC>(q, x + q.i + G.sz, hi);
    fwd_diff<T, VEC>(xc, hi, G.wz, t[2]);
  }
  T n2[VEC];
  for (int a = 0; a < G.ndim; ++a) {
    T wv[VEC];
    vzero(cc[a]);
    if (c) vlc<RAG, T, VEC>(q, c + a * G.n + q.i, cc[a]);
    vlc<RAG, T, VEC>(q, w + a * G.n + q.i, wv);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      t[a][k] = t[a][k] + wv[k] - cc[a][k];
      n2[k] = (a == 0) ? t[a][k] * t[a][k] : n2[k] + t[a][k] * t[a][k];
    }
  }
  T nrm[VEC], mag[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    nrm[k] = t_sqrt(n2[k]);
    mag[k] = t_max(t_abs(nrm[k]) - thr, T(0)) * t_sign(nrm[k]);
  }
  for (int a = 0; a < G.ndim; ++a) {
    T va[VEC], wa[VEC], ra[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      va[k] = (nrm[k] > thr) ? mag[k] * t[a][k] / nrm[k] : T(0);
      wa[k] = t[a][k] - va[k];
      ra[k] = rhs_scale * (va[k] - wa[k] + cc[a][k]);
      if (NORM && (!RAG || k < q.nval)) acc += (double)ra[k] * (double)ra[k];
    }
    if (v) vs<RAG, T, VEC>(q.nval, v + a * G.n + q.i, va);   // (v itself may be unwanted)
    vs<RAG, T, VEC>(q.nval, w + a * G.n + q.i, wa);
    if (rhs) vs<RAG, T, VEC>(q.nval, rhs + a * G.n + q.i, ra);
  }
  }
  if constexpr (NORM) {
    __shared__ double sred[kBlock / kWave];
    acc = wave_sum(acc);
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    if (lane == 0) sred[wv] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
      for (int k = 0; k < kBlock / kWave; ++k) t += sred[k];
      ws[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
    }
  }
}

// ---------------------------------------------------------------------------
// The outer step and the next solve's right-hand side vector in ONE pass (3-D, rows of
// whole vectors): k_admm_vw followed by k_lsmr_v (nsol_lsmr.hip; B = gradient, c_v = 0)
//   t = grad x + w - c,  v = shrink(t),  w' = t - v,  r = rhs_scale (v - w' + c)
//   g = c_atu atb + c_btu grad^T r
// without r ever going to memory (12 B per voxel written and 12 read again).  grad^T r
// at a voxel needs r_x of the voxel to its left, r_y of the one above and r_z of the one
// in the plane before; r at a point needs all three components of t there.  A workgroup
// owns 4 rows of 64 vectors and MARCHES along z, so r_z of the plane before is the lane's
// own value of the step before (registers), and x of the next plane, loaded for this
// step's forward difference, is the step after's own x.  r of the row above and of the
// ONE voxel to the left of the lane's vector are recomputed from x and w there (the
// arithmetic of k_admm_vw, so every r equals the value that kernel would have stored:
// w' and g are bit for bit those of the two kernels).  w is read at neighbouring points
// other lanes update, hence w_in / w_out (the caller alternates two arrays).
// A z chunk starts with one warm-up step on the plane before it (nothing stored).
// Sums: ws[block] = sum r^2 (own points), ws[nblocks + block] = sum g^2.
// ---------------------------------------------------------------------------
template <typename T, int N>
__device__ __forceinline__ void vw_point(const T (&gx)[N], const T (&gy)[N], const T (&gz)[N],
                                         const T (&w)[3][N], const T (&c)[3][N], T thr,
                                         T rs, T (&wn)[3][N], T (&r)[3][N]) {
  T t[3][N], n2[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    t[0][k] = gx[k] + w[0][k] - c[0][k];
    n2[k] = t[0][k] * t[0][k];
    t[1][k] = gy[k] + w[1][k] - c[1][k];
    n2[k] = n2[k] + t[1][k] * t[1][k];
    t[2][k] = gz[k] + w[2][k] - c[2][k];
    n2[k] = n2[k] + t[2][k] * t[2][k];
  }
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const T nrm = t_sqrt(n2[k]);
    const T mag = t_max(t_abs(nrm) - thr, T(0)) * t_sign(nrm);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const T va = (nrm > thr) ? mag * t[a][k] / nrm : T(0);
      wn[a][k] = t[a][k] - va;
      r[a][k] = rs * (va - wn[a][k] + c[a][k]);
    }
  }
}

// component A of r alone (the same operations as above for that component)
template <typename T, int N, int A>
__device__ __forceinline__ void vw_point_r(const T (&gx)[N], const T (&gy)[N],
                                           const T (&gz)[N], const T (&w)[3][N],
                                           const T (&c)[3][N], T thr, T rs, T (&r)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) {
    T t[3];
    t[0] = gx[k] + w[0][k] - c[0][k];
    T n2 = t[0] * t[0];
    t[1] = gy[k] + w[1][k] - c[1][k];
    n2 = n2 + t[1] * t[1];
    t[2] = gz[k] + w[2][k] - c[2][k];
    n2 = n2 + t[2] * t[2];
    const T nrm = t_sqrt(n2);
    const T mag = t_max(t_abs(nrm) - thr, T(0)) * t_sign(nrm);
    const T va = (nrm > thr) ? mag * t[A] / nrm : T(0);
    const T wn = t[A] - va;
    r[k] = rs * (va - wn + c[A][k]);
  }
}

template <typename T, int VEC, bool HASC>
__global__ __launch_bounds__(kBlock) void k_admm_vw_g(
    const T *__restrict__ x, const T *__restrict__ w_in, T *__restrict__ w_out,
    const T *__restrict__ c, const T *__restrict__ atb, T *__restrict__ g, Geom<T> G,
    int64_t zchunk, int nyg, T thr, T rs, T c_atu, T c_btu, double *ws, int64_t nblocks) {
  constexpr int XT = kBlock / 4;                      // lanes along x, 4 rows per workgroup
  // Which XCD runs a workgroup is its linear id (blockIdx.x + gridDim.x * blockIdx.y) mod 8:
  // with row groups in their natural order the group above a workgroup's rows runs on
  // ANOTHER XCD and everything the first row re-reads there is a second trip to memory.
  // The P values of blockIdx.y one XCD cycle spans stand for P slabs of consecutive row
  // groups instead (as voxel_at does for the stand-alone stencils).
  int yg = (int)(blockIdx.y % (unsigned)nyg);
  const int zc = (int)(blockIdx.y / (unsigned)nyg);
  {
    const unsigned gxd = gridDim.x;
    const unsigned P = gxd == 1 ? 8u : (gxd == 2 ? 4u : (gxd == 4 ? 2u : 1u));
    if (P > 1 && (unsigned)nyg % P == 0) {
      const unsigned per = (unsigned)nyg / P;
      yg = (int)(((unsigned)yg % P) * per + (unsigned)yg / P);
    }
  }
  const int64_t ix = ((int64_t)blockIdx.x * XT + (threadIdx.x % XT)) * VEC;
  const int64_t iy = (int64_t)yg * 4 + (threadIdx.x / XT);
  const bool ok = ix < G.nx && iy < G.ny;
  const int64_t zbeg = (int64_t)zc * zchunk;
  int64_t zend = zbeg + zchunk;
  if (zend > G.nz) zend = G.nz;
  const bool has_r = ix + VEC < G.nx, has_d = iy + 1 < G.ny, has_u = iy > 0, has_l = ix > 0;
  const int wl = threadIdx.x % kWave;                 // lane in the wave (= in the row)
  double acc_r = 0.0, acc_g = 0.0;
  if (ok) {
    auto ldv = [&](const T *p, int64_t i, T (&v)[VEC]) { vload<T, VEC>(p + i, v); };
    // state carried from plane to plane: x of this plane at the own point, the row above
    // and the voxel to the left; r_z of the plane before at the own point
    T xc[VEC], xu[VEC], rz_prev[VEC];
    T xl = T(0);
    vzero(rz_prev);
    const int64_t z0 = zbeg > 0 ? zbeg - 1 : 0;       // (warm-up plane)
    const int64_t row0 = iy * G.sy + ix;
    ldv(x, z0 * G.sz + row0, xc);
    vzero(xu);
    if (has_u) ldv(x, z0 * G.sz + row0 - G.sy, xu);
    if (has_l) xl = x[z0 * G.sz + row0 - 1];
    for (int64_t z = z0; z < zend; ++z) {
      const int64_t i = z * G.sz + row0;
      const bool has_n = z + 1 < G.nz;
      // x of the next plane (own, above, left), of the row below, to the right
      T xn[VEC], xun[VEC], xd[VEC];
      T xln = T(0);
      vzero(xn); vzero(xun); vzero(xd);
      if (has_n) {
        ldv(x, i + G.sz, xn);
        if (has_u) ldv(x, i + G.sz - G.sy, xun);
      }
      xln = __shfl_up(xn[VEC - 1], 1, kWave);
      if (wl == 0) xln = (has_n && has_l) ? x[i + G.sz - 1] : T(0);
      if (has_d) ldv(x, i + G.sy, xd);
      // (the voxel right of the lane's vector is the next lane's first one: only the
      // wave's last lane loads it; likewise the values left of the vector below)
      const T right_sh = __shfl_down(xc[0], 1, kWave);
      const T right = !has_r ? T(0) : (wl < kWave - 1 ? right_sh : x[i + VEC]);
      // ---- own point
      T gx[VEC], gy[VEC], gz[VEC], wv[3][VEC], cv[3][VEC], wn[3][VEC], r[3][VEC];
      fwd_diff_x<T, VEC>(xc, right, G.wx, gx);
      fwd_diff<T, VEC>(xc, xd, G.wy, gy);
      fwd_diff<T, VEC>(xc, xn, G.wz, gz);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        ldv(w_in, a * G.n + i, wv[a]);
        vzero(cv[a]);
        if (HASC) ldv(c, a * G.n + i, cv[a]);
      }
      vw_point<T, VEC>(gx, gy, gz, wv, cv, thr, rs, wn, r);
      // ---- the row above: r_y there
      T ru_y[VEC];
      vzero(ru_y);
      if (has_u) {                                    // (uniform in a wave: one row each)
        const int64_t iu = i - G.sy;
        const T ru_sh = __shfl_down(xu[0], 1, kWave);
        const T right_u = !has_r ? T(0) : (wl < kWave - 1 ? ru_sh : x[iu + VEC]);
        T ux[VEC], uy[VEC], uz[VEC], uw[3][VEC], uc[3][VEC];
        fwd_diff_x<T, VEC>(xu, right_u, G.wx, ux);
        fwd_diff<T, VEC>(xu, xc, G.wy, uy);
        fwd_diff<T, VEC>(xu, xun, G.wz, uz);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          ldv(w_in, a * G.n + iu, uw[a]);
          vzero(uc[a]);
          if (HASC) ldv(c, a * G.n + iu, uc[a]);
        }
        vw_point_r<T, VEC, 1>(ux, uy, uz, uw, uc, thr, rs, ru_y);
      }
      // ---- the voxel to the left: r_x there
      T rl_x = T(0);
      {
        const int64_t il = i - 1;
        T lx_[1], ly_[1], lz_[1], lw[3][1], lc[3][1], lr[1];
        const bool edge = wl == 0;                    // (the wave's first lane loads)
        T below = __shfl_up(xd[VEC - 1], 1, kWave);
        if (edge) below = (has_l && has_d) ? x[il + G.sy] : T(0);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          lw[a][0] = __shfl_up(wv[a][VEC - 1], 1, kWave);
          lc[a][0] = __shfl_up(cv[a][VEC - 1], 1, kWave);
          if (edge) {
            lw[a][0] = has_l ? w_in[a * G.n + il] : T(0);
            lc[a][0] = (HASC && has_l) ? c[a * G.n + il] : T(0);
          }
        }
        if (has_l) {
          lx_[0] = xc[0] * G.wx + xl * (-G.wx);
          ly_[0] = below * G.wy + xl * (-G.wy);
          lz_[0] = xln * G.wz + xl * (-G.wz);
          vw_point_r<T, 1, 0>(lx_, ly_, lz_, lw, lc, thr, rs, lr);
          rl_x = lr[0];
        }
      }
      if (z >= zbeg) {
        // ---- grad^T r and the new vector, k_lsmr_v's order
        T val[VEC], ab[VEC];
        ldv(atb, i, ab);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const T l = (k > 0) ? r[0][(k + VEC - 1) % VEC] : rl_x;
          T kt = r[0][k] * (-G.wx) + l * G.wx;
          kt += r[1][k] * (-G.wy) + ru_y[k] * G.wy;
          kt += r[2][k] * (-G.wz) + rz_prev[k] * G.wz;
          val[k] = c_atu * ab[k];
          val[k] += c_btu * kt;
          val[k] += T(0) * ab[k];
          acc_g += (double)val[k] * (double)val[k];
#pragma unroll
          for (int a = 0; a < 3; ++a) acc_r += (double)r[a][k] * (double)r[a][k];
        }
        vstore<T, VEC>(g + i, val);
#pragma unroll
        for (int a = 0; a < 3; ++a) vstore<T, VEC>(w_out + a * G.n + i, wn[a]);
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        rz_prev[k] = r[2][k];
        xc[k] = xn[k];
        xu[k] = xun[k];
      }
      xl = xln;
    }
  }
  __shared__ double sred[2][kBlock / kWave];
  acc_r = wave_sum(acc_r);
  acc_g = wave_sum(acc_g);
  const int lane = threadIdx.x & (kWave - 1), wvi = threadIdx.x / kWave;
  if (lane == 0) { sred[0][wvi] = acc_r; sred[1][wvi] = acc_g; }
  __syncthreads();
  if (threadIdx.x < 2) {
    double t = 0.0;
    for (int k = 0; k < kBlock / kWave; ++k) t += sred[threadIdx.x][k];
    ws[(int64_t)threadIdx.x * nblocks + (int64_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
  }
}

// result[b] = the sum of the b-th run of n partials, fixed order
__global__ __launch_bounds__(kBlock) void k_reduce_final2(const double *ws, int64_t n,
                                                           double *result) {
  ws += (int64_t)blockIdx.x * n;
  double v = 0.0;
  for (int64_t k = threadIdx.x; k < n; k += kBlock) v += ws[k];
  __shared__ double s[kBlock / kWave];
  v = wave_sum(v);
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  if (lane == 0) s[wv] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < kBlock / kWave; ++k) t += s[k];
    result[blockIdx.x] = t;
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_vector_shrink(const T *__restrict__ t,
                                                           T *__restrict__ v,
                                                           int ndim, int64_t m,
                                                           T thr) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m;
       i += stride) {
    T tt[3];
    T n2 = T(0);
    for (int a = 0; a < ndim; ++a) {
      tt[a] = t[a * m + i];
      n2 = (a == 0) ? tt[a] * tt[a] : n2 + tt[a] * tt[a];
    }
    const T nrm = t_sqrt(n2);
    const bool on = nrm > thr;
    const T mag = t_max(t_abs(nrm) - thr, T(0)) * t_sign(nrm);
    for (int a = 0; a < ndim; ++a) v[a * m + i] = on ? mag * tt[a] / nrm : T(0);
  }
}

// ---------------------------------------------------------- robust loss ----
// loss_functions.py:82-248: rho(z*s2)... evaluated on f2 = r^2, z = f2 / f_scale^2
template <typename T>
__device__ __forceinline__ void loss_eval(int loss, T f2, T s2, T &rho, T &drho,
                                          T gm = T(1.345)) {
  const T z = f2 / s2;
  switch (loss) {
    case NSOL_LOSS_SOFT_L1: {
      const T q = t_sqrt(T(1) + z);
      rho = T(2) * (q - T(1)) * s2;
      drho = T(1) / q;
    } break;
    case NSOL_LOSS_HUBER: {
      const T g2 = gm * gm;
      if (z < g2) { rho = z * s2; drho = T(1); }
      else {
        const T q = t_sqrt(z);
        rho = (T(2) * gm * q - g2) * s2;
        drho = gm / q;
      }
    } break;
    case NSOL_LOSS_CAUCHY:
      rho = (T)log1p((double)z) * s2;
      drho = T(1) / (T(1) + z);
      break;
    case NSOL_LOSS_ARCTAN:
      rho = (T)atan((double)z) * s2;
      drho = T(1) / (T(1) + z * z);
      break;
    default:
      rho = f2;
      drho = T(1);
  }
}

// residual r (minus b when given, as nsol_lincomb2 forms A x - b), its loss and
// rho'(r^2) * r; VEC elements = 16 bytes per lane and trip when the arrays allow
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_loss(const T *r, const T *b, T *g,
                                                  int64_t n, int loss, T s2,
                                                  double *ws) {
  typedef T V __attribute__((ext_vector_type(VEC)));
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t nv = n / VEC;
  double acc = 0.0;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nv;
       j += stride) {
    V rv = reinterpret_cast<const V *>(r)[j];
    if (b) {
      const V bv = reinterpret_cast<const V *>(b)[j];
#pragma unroll
      for (int e = 0; e < VEC; ++e) rv[e] = T(1) * rv[e] + T(-1) * bv[e];
    }
    V gv;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      T rho, drho;
      loss_eval(loss, rv[e] * rv[e], s2, rho, drho);
      acc += (double)rho;
      gv[e] = drho * rv[e];
    }
    if (g) reinterpret_cast<V *>(g)[j] = gv;
  }
  block_store_partial(acc, ws);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_loss_eval(const T *__restrict__ f2,
                                                       T *rho, T *drho,
                                                       int64_t n, int loss,
                                                       T s2, T gm) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += stride) {
    T r, d;
    loss_eval(loss, f2[i], s2, r, d, gm);
    if (rho) rho[i] = r;
    if (drho) drho[i] = d;
  }
}

// prior_measures.py:27-52
template <typename T>
__global__ __launch_bounds__(kBlock) void k_vector_norm_sum(
    const T *__restrict__ t, int ndim, int64_t m, int mode, T gm, double *ws) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m;
       i += stride) {
    T n2 = T(0);
    for (int a = 0; a < ndim; ++a) {
      const T v = t[a * m + i];
      n2 = (a == 0) ? v * v : n2 + v * v;
    }
    if (mode == 0) {
      acc += (double)t_sqrt(n2);
    } else {
      T r, d;
      loss_eval(NSOL_LOSS_HUBER, n2, T(1), r, d, gm);
      acc += (double)(r / (T(2) * gm));
    }
  }
  block_store_partial(acc, ws);
}

// similarity_measures.py:26-120: all sums of one (x, x_ref) pair in one pass
constexpr int kPairStats = 8;
constexpr int kPairBlocks = 2048;

template <typename T>
__global__ __launch_bounds__(kBlock) void k_pair_stats(const T *__restrict__ x,
                                                        const T *__restrict__ y,
                                                        int64_t n, double mx,
                                                        double my, double *ws) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double a[kPairStats] = {0, 0, 0, 0, 0, -1.7976931348623157e308, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += stride) {
    const double xv = (double)x[i], yv = (double)y[i];
    const double dx = xv - mx, dy = yv - my, d = xv - yv;
    a[0] += dx * dy; a[1] += dx * dx; a[2] += dy * dy;
    a[3] += fabs(d); a[4] += d * d;
    a[5] = fmax(a[5], yv);
    a[6] += xv; a[7] += yv;
  }
  __shared__ double s[kPairStats][kBlock / kWave];
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
#pragma unroll
  for (int k = 0; k < kPairStats; ++k) {
    double v = a[k];
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
      const double o = __shfl_down(v, off, kWave);
      v = (k == 5) ? fmax(v, o) : v + o;
    }
    if (lane == 0) s[k][wv] = v;
  }
  __syncthreads();
  if (threadIdx.x < kPairStats) {
    const int k = threadIdx.x;
    double t = s[k][0];
    for (int j = 1; j < kBlock / kWave; ++j)
      t = (k == 5) ? fmax(t, s[k][j]) : t + s[k][j];
    ws[(int64_t)k * kPairBlocks + blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(kBlock) void k_pair_final(const double *ws,
                                                        int nparts,
                                                        double *result) {
  // one wave per statistic would do; keep it simple: thread k sums row k
  if (threadIdx.x < kPairStats) {
    const int k = threadIdx.x;
    double t = ws[(int64_t)k * kPairBlocks];
    for (int j = 1; j < nparts; ++j) {
      const double v = ws[(int64_t)k * kPairBlocks + j];
      t = (k == 5) ? fmax(t, v) : t + v;
    }
    result[k] = t;
  }
}

template <typename T>
int pair_stats_impl(const T *x, const T *y, int64_t n, double mx, double my,
                    double *result, double *ws, void *stream) {
  if (n < 1 || !x || !y || !result || !ws) return NSOL_EINVAL;
  static_assert(kPairStats * kPairBlocks <= kReducePartials, "workspace");
  int g = grid_for(n);
  if (g > kPairBlocks) g = kPairBlocks;
  hipLaunchKernelGGL(k_pair_stats<T>, dim3(g), dim3(kBlock), 0, as_stream(stream),
                     x, y, n, mx, my, ws);
  hipLaunchKernelGGL(k_pair_final, dim3(1), dim3(kBlock), 0, as_stream(stream),
                     ws, g, result);
  return launch_status();
}

// ------------------------------------------------------- typed front ends ----
template <typename T>
int loss_eval_impl(const T *f2, T *rho, T *drho, int64_t n, int loss,
                   double f_scale, double gm, void *stream) {
  if (n < 0 || !f2 || loss < 0 || loss > 4) return NSOL_EINVAL;
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_loss_eval<T>, dim3(grid_for(n)), dim3(kBlock), 0,
                     as_stream(stream), f2, rho, drho, n, loss,
                     (T)(f_scale * f_scale), (T)gm);
  return launch_status();
}

template <typename T>
int vector_norm_sum_impl(const T *t, int ndim, int64_t m, int mode, double gm,
                         double *result, double *ws, void *stream) {
  if (ndim < 1 || ndim > 3 || m < 0 || !t || !result || !ws || mode < 0 ||
      mode > 1)
    return NSOL_EINVAL;
  const int gr = reduce_grid(m);
  hipLaunchKernelGGL(k_vector_norm_sum<T>, dim3(gr), dim3(kBlock), 0,
                     as_stream(stream), t, ndim, m, mode, (T)gm, ws);
  hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(kBlock), 0, as_stream(stream),
                     ws, gr, result, 1.0);
  return launch_status();
}

template <typename T>
int grad_impl(const T *x, T *g, int ndim, int64_t nz, int64_t ny, int64_t nx,
              double wx, double wy, double wz, void *stream) {
  NSOL_CHECK_GEOM(ndim, nz, ny, nx);
  if (!x || !g) return NSOL_EINVAL;
  Geom<T> G = make_geom<T>(ndim, nz, ny, nx, wx, wy, wz);
  return dispatch_stencil<T>(nz, ny, nx, ptr16(x) && ptr16(g) && G.n % 4 == 0,
                             [&](auto vec, auto rows, auto rag) {
    constexpr int V = decltype(vec)::value, R = decltype(rows)::value;
    constexpr bool RG = decltype(rag)::value;
    hipLaunchKernelGGL((k_grad<T, V, R, RG>), (stencil_grid<V, R>(nz, ny, nx)),
                       dim3(kBlock), 0, as_stream(stream), x, g, G);
    return launch_status();
  });
}

template <typename T>
int grad_adj_impl(const T *p, T *out, int ndim, int64_t nz, int64_t ny,
                  int64_t nx, double wx, double wy, double wz, void *stream) {
  NSOL_CHECK_GEOM(ndim, nz, ny, nx);
  if (!p || !out) return NSOL_EINVAL;
  Geom<T> G = make_geom<T>(ndim, nz, ny, nx, wx, wy, wz);
  return dispatch_stencil<T>(nz, ny, nx, ptr16(p) && ptr16(out) && G.n % 4 == 0,
                             [&](auto vec, auto rows, auto rag) {
    constexpr int V = decltype(vec)::value, R = decltype(rows)::value;
    constexpr bool RG = decltype(rag)::value;
    hipLaunchKernelGGL((k_grad_adj<T, V, R, RG>), (stencil_grid<V, R>(nz, ny, nx)),
                       dim3(kBlock), 0, as_stream(stream), p, out, G);
    return launch_status();
  });
}

template <typename T>
int grad_adj_axpy_impl(const T *p, const T *x, T *out, int ndim, int64_t nz, int64_t ny,
                       int64_t nx, double wx, double wy, double wz, double tau,
                       void *stream) {
  NSOL_CHECK_GEOM(ndim, nz, ny, nx);
  if (!p || !x || !out) return NSOL_EINVAL;
  Geom<T> G = make_geom<T>(ndim, nz, ny, nx, wx, wy, wz);
  return dispatch_stencil<T>(nz, ny, nx,
                             ptr16(p) && ptr16(x) && ptr16(out) && G.n % 4 == 0,
                             [&](auto vec, auto rows, auto rag) {
    constexpr int V = decltype(vec)::value, R = decltype(rows)::value;
    constexpr bool RG = decltype(rag)::value;
    hipLaunchKernelGGL((k_grad_adj<T, V, R, RG, true>), (stencil_grid<V, R>(nz, ny, nx)),
                       dim3(kBlock), 0, as_stream(stream), p, out, G, x, (T)tau);
    return launch_status();
  });
}

template <typename T>
int diff_axis_impl(const T *x, T *out, int dir, int adjoint, int64_t nz,
                   int64_t ny, int64_t nx, double w, void *stream) {
  if (dir < 0 || dir > 2 || nz < 1 || ny < 1 || nx < 1 || !x || !out || x == out)
    return NSOL_EINVAL;
  Geom<T> G = make_geom<T>(3, nz, ny, nx, 1, 1, 1);
  hipLaunchKernelGGL(k_diff_axis<T>, dim3(grid_for(G.n)), dim3(kBlock), 0,
                     as_stream(stream), x, out, G, dir, adjoint, (T)w);
  return launch_status();
}

template <typename T>
int dot_impl(const T *x, const T *y, int64_t n, double *result, double *ws,
             void *stream) {
  if (n < 0 || !result || !ws || (n > 0 && (!x || !y))) return NSOL_EINVAL;
  const int g = reduce_grid(n);
  hipLaunchKernelGGL(k_dot<T>, dim3(g), dim3(kBlock), 0, as_stream(stream), x, y,
                     n, ws);
  hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(kBlock), 0, as_stream(stream),
                     ws, g, result, 1.0);
  return launch_status();
}

template <typename T>
int admm_vw_impl(const T *x, T *v, T *w, const T *c, T *rhs, int ndim,
                 int64_t nz, int64_t ny, int64_t nx, double wx, double wy,
                 double wz, double thr, double rhs_scale, void *stream,
                 double *result = nullptr, double *ws = nullptr) {
  NSOL_CHECK_GEOM(ndim, nz, ny, nx);
  if (!x || !w || (result && (!ws || !rhs))) return NSOL_EINVAL;
  Geom<T> G = make_geom<T>(ndim, nz, ny, nx, wx, wy, wz);
  const bool al = ptr16(x) && (!v || ptr16(v)) && ptr16(w) && (!c || ptr16(c)) &&
                  (!rhs || ptr16(rhs)) && G.n % 4 == 0;
  return dispatch_stencil<T>(nz, ny, nx, al, [&](auto vec, auto rows, auto rag) {
    constexpr int V = decltype(vec)::value, R = decltype(rows)::value;
    constexpr bool RG = decltype(rag)::value;
    const dim3 grid = stencil_grid<V, R>(nz, ny, nx);
    if (result) {
      const int nparts = (int)(grid.x * grid.y);
      if (nparts > kReducePartials) return NSOL_EINVAL;
      hipLaunchKernelGGL((k_admm_vw<T, V, R, RG, true>), grid, dim3(kBlock), 0,
                         as_stream(stream), x, v, w, c, rhs, G, (T)thr, (T)rhs_scale, ws);
      hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(kBlock), 0, as_stream(stream), ws,
                         nparts, result, 1.0);
      return launch_status();
    }
    hipLaunchKernelGGL((k_admm_vw<T, V, R, RG>), grid, dim3(kBlock), 0,
                       as_stream(stream), x, v, w, c, rhs, G, (T)thr, (T)rhs_scale,
                       (double *)nullptr);
    return launch_status();
  });
}

// -2: the one-pass form does not apply (nothing launched): nsol_admm_vw_update_norm_*
// and nsol_lsmr_v_update_to_* then
template <typename T>
int admm_vw_g_impl(const T *x, const T *w_in, T *w_out, const T *c, const T *atb, T *g,
                   int ndim, int64_t nz, int64_t ny, int64_t nx, double wx, double wy,
                   double wz, double thr, double rhs_scale, double c_atu, double c_btu,
                   double *result, double *ws, void *stream) {
  NSOL_CHECK_GEOM(ndim, nz, ny, nx);
  if (!x || !w_in || !w_out || w_in == w_out || !atb || !g || g == atb || g == x ||
      !result || !ws)
    return NSOL_EINVAL;
  constexpr int VEC = 16 / sizeof(T);
  if (ndim != 3 || nx % VEC != 0 || nx < VEC || !ptr16(x) || !ptr16(w_in) || !ptr16(w_out) ||
      (c && !ptr16(c)) || !ptr16(atb) || !ptr16(g) || (nz * ny * nx) % 4 != 0)
    return -2;
  const Geom<T> G = make_geom<T>(ndim, nz, ny, nx, wx, wy, wz);
  constexpr int XT = kBlock / 4;
  const int64_t gx = (nx + (int64_t)XT * VEC - 1) / ((int64_t)XT * VEC);
  const int64_t nyg = (ny + 3) / 4;
  // z chunks: enough workgroups to fill the chip a few times over, chunks long enough
  // for the warm-up plane not to matter
  // (512^3: 1 024 workgroups 1.26 ms, 2 048 ... 16 384 within 2 % of each other; 512 1.48)
  int64_t chunks = (8 * 256 + gx * nyg - 1) / (gx * nyg);
  if (chunks < 1) chunks = 1;
  int64_t zchunk = (nz + chunks - 1) / chunks;
  if (zchunk < 16) zchunk = nz < 16 ? nz : 16;
  chunks = (nz + zchunk - 1) / zchunk;
  const int64_t nblocks = gx * nyg * chunks;
  if (gx > 65535 || nyg * chunks > 65535 || 2 * nblocks > kReducePartials) return -2;
  const dim3 grid((unsigned)gx, (unsigned)(nyg * chunks), 1);
  if (c)
    hipLaunchKernelGGL((k_admm_vw_g<T, VEC, true>), grid, dim3(kBlock), 0, as_stream(stream),
                       x, w_in, w_out, c, atb, g, G, zchunk, (int)nyg, (T)thr, (T)rhs_scale,
                       (T)c_atu, (T)c_btu, ws, nblocks);
  else
    hipLaunchKernelGGL((k_admm_vw_g<T, VEC, false>), grid, dim3(kBlock), 0, as_stream(stream),
                       x, w_in, w_out, c, atb, g, G, zchunk, (int)nyg, (T)thr, (T)rhs_scale,
                       (T)c_atu, (T)c_btu, ws, nblocks);
  hipLaunchKernelGGL(k_reduce_final2, dim3(2), dim3(kBlock), 0, as_stream(stream), ws, nblocks,
                     result);
  return launch_status();
}

template <typename T>
int shrink_impl(const T *t, T *v, int ndim, int64_t m, double thr, void *stream) {
  if (ndim < 1 || ndim > 3 || m < 0 || !t || !v) return NSOL_EINVAL;
  if (m == 0) return 0;
  hipLaunchKernelGGL(k_vector_shrink<T>, dim3(grid_for(m)), dim3(kBlock), 0,
                     as_stream(stream), t, v, ndim, m, (T)thr);
  return launch_status();
}

template <typename T>
int loss_impl(const T *r, const T *b, T *g, int64_t n, int loss, double f_scale,
              double *result, double *ws, void *stream) {
  if (n < 0 || !r || !result || !ws || loss < 0 || loss > 4) return NSOL_EINVAL;
  constexpr int VW = 16 / sizeof(T);
  const bool vec = n % VW == 0 &&
                   !((reinterpret_cast<uintptr_t>(r) | reinterpret_cast<uintptr_t>(b) |
                      reinterpret_cast<uintptr_t>(g)) & 15);
  const int gr = reduce_grid(vec ? n / VW : n);
  const T s2 = (T)(f_scale * f_scale);
  if (vec)
    hipLaunchKernelGGL((k_loss<T, VW>), dim3(gr), dim3(kBlock), 0, as_stream(stream), r,
                       b, g, n, loss, s2, ws);
  else
    hipLaunchKernelGGL((k_loss<T, 1>), dim3(gr), dim3(kBlock), 0, as_stream(stream), r,
                       b, g, n, loss, s2, ws);
  hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(kBlock), 0, as_stream(stream),
                     ws, gr, result, 0.5);
  return launch_status();
}

}  // namespace

// ================================ C ABI ====================================
extern "C" {

int nsol_hip_abi_version(void) { return NSOL_HIP_ABI_VERSION; }
int nsol_hip_reduce_ws_doubles(void) { return kReducePartials; }

#define NSOL_DEF2(NAME, T, SUF)                                                  \
  int nsol_grad_##SUF(const T *x, T *g, int ndim, int64_t nz, int64_t ny,        \
                      int64_t nx, double wx, double wy, double wz, void *s) {    \
    return grad_impl<T>(x, g, ndim, nz, ny, nx, wx, wy, wz, s);                  \
  }                                                                              \
  int nsol_grad_adj_##SUF(const T *p, T *o, int ndim, int64_t nz, int64_t ny,    \
                          int64_t nx, double wx, double wy, double wz,           \
                          void *s) {                                             \
    return grad_adj_impl<T>(p, o, ndim, nz, ny, nx, wx, wy, wz, s);              \
  }                                                                              \
  int nsol_grad_adj_axpy_##SUF(const T *p, const T *x, T *o, int ndim,           \
                               int64_t nz, int64_t ny, int64_t nx, double wx,    \
                               double wy, double wz, double tau, void *s) {      \
    return grad_adj_axpy_impl<T>(p, x, o, ndim, nz, ny, nx, wx, wy, wz, tau, s); \
  }                                                                              \
  int nsol_extrapolate_##SUF(T *out, const T *a, const T *b, double theta,       \
                             int64_t n, void *s) {                               \
    if (n > 0 && (!out || !a || !b)) return NSOL_EINVAL;                         \
    const T th = (T)theta;                                                       \
    return launch_map<T>(n, s, [=] __device__(int64_t i) {                       \
      const T v = a[i];                                                          \
      out[i] = v + th * (v - b[i]);                                              \
    });                                                                          \
  }                                                                              \
  int nsol_diff_axis_##SUF(const T *x, T *o, int dir, int adj, int64_t nz,       \
                           int64_t ny, int64_t nx, double w, void *s) {          \
    return diff_axis_impl<T>(x, o, dir, adj, nz, ny, nx, w, s);                  \
  }                                                                              \
  int nsol_lincomb2_##SUF(T *out, double a, const T *x, double b, const T *y,    \
                          int64_t n, void *s) {                                  \
    if (n > 0 && (!out || !x || !y)) return NSOL_EINVAL;                         \
    const T ta = (T)a, tb = (T)b;                                                \
    return launch_map<T>(n, s, [=] __device__(int64_t i) {                       \
      out[i] = ta * x[i] + tb * y[i];                                            \
    });                                                                          \
  }                                                                              \
  int nsol_lincomb3_##SUF(T *out, double a, const T *x, double b, const T *y,    \
                          double c, const T *z, int64_t n, void *s) {            \
    if (n > 0 && (!out || !x || !y || !z)) return NSOL_EINVAL;                   \
    const T ta = (T)a, tb = (T)b, tc = (T)c;                                     \
    return launch_map<T>(n, s, [=] __device__(int64_t i) {                       \
      out[i] = ta * x[i] + tb * y[i] + tc * z[i];                                \
    });                                                                          \
  }                                                                              \
  int nsol_scale_##SUF(T *out, const T *x, double a, int divide, int64_t n,      \
                       void *s) {                                                \
    if (n > 0 && (!out || !x)) return NSOL_EINVAL;                               \
    const T ta = (T)a;                                                           \
    if (divide)                                                                  \
      return launch_map<T>(n, s,                                                 \
                           [=] __device__(int64_t i) { out[i] = x[i] / ta; });   \
    return launch_map<T>(n, s,                                                   \
                         [=] __device__(int64_t i) { out[i] = x[i] * ta; });     \
  }                                                                              \
  int nsol_clip_##SUF(T *out, const T *x, double lo, double hi, int64_t n,       \
                      void *s) {                                                 \
    if (n > 0 && (!out || !x)) return NSOL_EINVAL;                               \
    const T tlo = (T)lo, thi = (T)hi;                                            \
    return launch_map<T>(n, s, [=] __device__(int64_t i) {                       \
      T v = x[i];                                                                \
      v = v < tlo ? tlo : v;                                                     \
      out[i] = v > thi ? thi : v;                                                \
    });                                                                          \
  }                                                                              \
  int nsol_prox_dual_clamp_##SUF(T *out, const T *x, double den, int64_t n,      \
                                 void *s) {                                      \
    if (n > 0 && (!out || !x)) return NSOL_EINVAL;                               \
    const T td = huber_den<T>(den);                                              \
    return launch_map<T>(n, s, [=] __device__(int64_t i) {                       \
      out[i] = dual_clamp<T>(huber_div<T>(x[i], td));                            \
    });                                                                          \
  }                                                                              \
  int nsol_prox_ell2_##SUF(T *out, const T *x, const T *bt, double tau,          \
                           int64_t n, void *s) {                                 \
    if (n > 0 && (!out || !x || !bt)) return NSOL_EINVAL;                        \
    const T tl = (T)tau, opt = prox_den<T>(tau);                                   \
    return launch_map<T>(n, s, [=] __device__(int64_t i) {                       \
      out[i] = prox_data<T>(x[i], bt[i], tl, opt, false);                        \
    });                                                                          \
  }                                                                              \
  int nsol_prox_ell1_##SUF(T *out, const T *x, const T *bt, double tau,          \
                           int64_t n, void *s) {                                 \
    if (n > 0 && (!out || !x || !bt)) return NSOL_EINVAL;                        \
    const T tl = (T)tau;                                                         \
    return launch_map<T>(n, s, [=] __device__(int64_t i) {                       \
      out[i] = prox_data<T>(x[i], bt[i], tl, T(1), true);                        \
    });                                                                          \
  }                                                                              \
  int nsol_dot_##SUF(const T *x, const T *y, int64_t n, double *r, double *ws,   \
                     void *s) {                                                  \
    return dot_impl<T>(x, y, n, r, ws, s);                                       \
  }                                                                              \
  int nsol_admm_vw_update_##SUF(const T *x, T *v, T *w, const T *c, T *rhs,      \
                                int ndim, int64_t nz, int64_t ny, int64_t nx,    \
                                double wx, double wy, double wz, double thr,     \
                                double rs, void *s) {                            \
    return admm_vw_impl<T>(x, v, w, c, rhs, ndim, nz, ny, nx, wx, wy, wz, thr,   \
                           rs, s);                                               \
  }                                                                              \
  int nsol_admm_vw_update_norm_##SUF(const T *x, T *v, T *w, const T *c, T *rhs, \
                                     int ndim, int64_t nz, int64_t ny,           \
                                     int64_t nx, double wx, double wy, double wz,\
                                     double thr, double rs, double *res,         \
                                     double *ws, void *s) {                      \
    if (!res || !ws) return NSOL_EINVAL;                                         \
    return admm_vw_impl<T>(x, v, w, c, rhs, ndim, nz, ny, nx, wx, wy, wz, thr,   \
                           rs, s, res, ws);                                      \
  }                                                                              \
  int nsol_admm_vw_update_g_##SUF(const T *x, const T *w_in, T *w_out, const T *c, \
                                  const T *atb, T *g, int ndim, int64_t nz,      \
                                  int64_t ny, int64_t nx, double wx, double wy,  \
                                  double wz, double thr, double rs, double c_atu, \
                                  double c_btu, double *res, double *ws, void *s) { \
    return admm_vw_g_impl<T>(x, w_in, w_out, c, atb, g, ndim, nz, ny, nx, wx, wy, \
                             wz, thr, rs, c_atu, c_btu, res, ws, s);             \
  }                                                                              \
  int nsol_vector_shrink_##SUF(const T *t, T *v, int ndim, int64_t m,            \
                               double thr, void *s) {                            \
    return shrink_impl<T>(t, v, ndim, m, thr, s);                                \
  }                                                                              \
  int nsol_loss_cost_grad_##SUF(const T *r, T *g, int64_t n, int loss,           \
                                double fs, double *res, double *ws, void *s) {   \
    return loss_impl<T>(r, nullptr, g, n, loss, fs, res, ws, s);                 \
  }                                                                              \
  int nsol_loss_residual_cost_grad_##SUF(const T *ax, const T *b, T *g,          \
                                         int64_t n, int loss, double fs,         \
                                         double *res, double *ws, void *s) {     \
    if (!b) return NSOL_EINVAL;                                                  \
    return loss_impl<T>(ax, b, g, n, loss, fs, res, ws, s);                      \
  }                                                                              \
  int nsol_loss_eval_##SUF(const T *f2, T *rho, T *drho, int64_t n, int loss,    \
                           double fs, double gm, void *s) {                      \
    return loss_eval_impl<T>(f2, rho, drho, n, loss, fs, gm, s);                 \
  }                                                                              \
  int nsol_vector_norm_sum_##SUF(const T *t, int ndim, int64_t m, int mode,      \
                                 double gm, double *res, double *ws, void *s) {  \
    return vector_norm_sum_impl<T>(t, ndim, m, mode, gm, res, ws, s);            \
  }                                                                              \
  int nsol_pair_stats_##SUF(const T *x, const T *y, int64_t n, double mx,        \
                            double my, double *res, double *ws, void *s) {       \
    return pair_stats_impl<T>(x, y, n, mx, my, res, ws, s);                      \
  }

NSOL_DEF2(_, float, f32)
NSOL_DEF2(_, double, f64)

}  // extern "C"
