// Fused Golub-Kahan / LSMR vector kernels for the augmented Tikhonov system
//   [A; sqrt(alpha) B] x = [b; sqrt(alpha) b_reg]
// (reference: tikhonov_linear_solver.py:226-274 + scipy lsmr.py:320-413).
// Each kernel makes ONE pass, applies B = grad / B = identity on the fly and
// leaves the squared norm of its output as deterministic fp64 partial sums, so
// one LSMR iteration costs 3 launches + the blur passes of A and A^T instead
// of ~14 axpy/scale/dot launches.  The bidiagonalisation vectors are kept
// UNNORMALISED in memory (u~ = beta*u, v~ = alpha*v); the 1/beta, 1/alpha
// factors ride in the coefficients of the consuming kernel.
#include <type_traits>

#include "nsol_common.hpp"
#include "nsol_stencil.hpp"

using namespace nsol;

namespace {

constexpr int kBNone = 0, kBGrad = 1, kBIdentity = 2;

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  return v;
}

__device__ __forceinline__ double wave_max_d(double v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, kWave));
  return v;
}

__device__ __forceinline__ void store_partial(double v, double *ws) {
  __shared__ double s[kBlock / kWave];
  v = wave_sum_d(v);
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  if (lane == 0) s[wv] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < kBlock / kWave; ++k) t += s[k];
    ws[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(kBlock) void k_final(const double *ws, int nparts,
                                                   double *result) {
  double v = 0.0;
  for (int k = threadIdx.x; k < nparts; k += kBlock) v += ws[k];
  __shared__ double s[kBlock / kWave];
  v = wave_sum_d(v);
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  if (lane == 0) s[wv] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < kBlock / kWave; ++k) t += s[k];
    result[0] = t;
  }
}

// partial sums land in ws[linear block id]
__device__ __forceinline__ void store_partial3(double v, double *ws) {
  __shared__ double s[kBlock / kWave];
  v = wave_sum_d(v);
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  if (lane == 0) s[wv] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < kBlock / kWave; ++k) t += s[k];
    ws[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
  }
}

// sums nblocks partials in a fixed order (deterministic)
__global__ __launch_bounds__(1024) void k_final_big(const double *ws,
                                                     int64_t nparts,
                                                     double *result) {
  double v = 0.0;
  for (int64_t k = threadIdx.x; k < nparts; k += 1024) v += ws[k];
  __shared__ double s[16];
  v = wave_sum_d(v);
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  if (lane == 0) s[wv] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < 16; ++k) t += s[k];
    result[0] = t;
  }
}

// u_top = c_av*Av + c_u*u_top ; u_bot = c_bv*B(v) + c_u*u_bot ; sum of squares
template <typename T, int VEC, int ROWS, bool RAG>
__global__ __launch_bounds__(kBlock) void k_lsmr_u(
    const T *__restrict__ Av, const T *__restrict__ v, T *__restrict__ u_top,
    T *__restrict__ u_bot, Geom<T> G, int bmode, T c_av, T c_bv, T c_u,
    double *ws) {
  const int64_t nrg = row_groups<T, VEC, ROWS>(G);
  double acc = 0.0;
  for (int64_t rg = blockIdx.y; rg < nrg; rg += gridDim.y) {
    const Voxel c = voxel_at<T, VEC, ROWS>(G, rg);
    if (!c.ok) continue;
    {
    T a[VEC], u[VEC];
    if (Av) {      // (null: the top block was updated elsewhere -- the blur's epilogue)
      vlc<RAG, T, VEC>(c, Av + c.i, a);
      vlc<RAG, T, VEC>(c, u_top + c.i, u);
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        u[k] = c_av * a[k] + c_u * u[k];
        if (!RAG || k < c.nval) acc += (double)u[k] * (double)u[k];
      }
      vs<RAG, T, VEC>(c.nval, u_top + c.i, u);
    }
    if (bmode == kBIdentity) {
      vlc<RAG, T, VEC>(c, v + c.i, a);
      vlc<RAG, T, VEC>(c, u_bot + c.i, u);
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        u[k] = c_bv * a[k] + c_u * u[k];
        if (!RAG || k < c.nval) acc += (double)u[k] * (double)u[k];
      }
      vs<RAG, T, VEC>(c.nval, u_bot + c.i, u);
    } else if (bmode == kBGrad) {
      T vc[VEC], hi[VEC], d[VEC];
      vlc<RAG, T, VEC>(c, v + c.i, vc);
      for (int dir = 0; dir < G.ndim; ++dir) {
        if (dir == 0) {
          const T right = (c.ix + VEC < G.nx) ? v[c.i + VEC] : T(0);
          fwd_diff_x<T, VEC>(vc, right, G.wx, d);
        } else if (dir == 1) {
          vzero(hi);
          if (c.iy + 1 < G.ny) vlc<RAG, T, VEC>(c, v + c.i + G.sy, hi);
          fwd_diff<T, VEC>(vc, hi, G.wy, d);
        } else {
          vzero(hi);
          if (c.iz + 1 < G.nz) vlc<RAG, T, VEC>(c, v + c.i + G.sz, hi);
          fwd_diff<T, VEC>(vc, hi, G.wz, d);
        }
        vlc<RAG, T, VEC>(c, u_bot + dir * G.n + c.i, u);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          u[k] = c_bv * d[k] + c_u * u[k];
          if (!RAG || k < c.nval) acc += (double)u[k] * (double)u[k];
        }
        vs<RAG, T, VEC>(c.nval, u_bot + dir * G.n + c.i, u);
      }
    }
    }
  }
  store_partial3(acc, ws);
}

// v = c_atu*Atu + c_btu*B^T(u_bot) + c_v*v ; sum of squares
template <typename T, int VEC, int ROWS, bool RAG>
__global__ __launch_bounds__(kBlock) void k_lsmr_v(
    const T *Atu, const T *__restrict__ u_bot, const T *v, T *v_out,
    Geom<T> G, int bmode, T c_atu, T c_btu, T c_v, double *ws) {
  // (v_out may be v -- the update in place -- or Atu: the new vector then takes the
  // place of A^T u and the old one stays, for a caller that keeps every v_k; each
  // value is read before it is written, by the same lane)
  const int64_t nrg = row_groups<T, VEC, ROWS>(G);
  double acc = 0.0;
  for (int64_t rg = blockIdx.y; rg < nrg; rg += gridDim.y) {
    const Voxel c = voxel_at<T, VEC, ROWS>(G, rg);
    if (!c.ok) continue;
    {
    T val[VEC], t[VEC], lo[VEC];
    vlc<RAG, T, VEC>(c, Atu + c.i, t);
#pragma unroll
    for (int k = 0; k < VEC; ++k) val[k] = c_atu * t[k];
    if (bmode == kBIdentity) {
      vlc<RAG, T, VEC>(c, u_bot + c.i, t);
#pragma unroll
      for (int k = 0; k < VEC; ++k) val[k] += c_btu * t[k];
    } else if (bmode == kBGrad) {
      T kt[VEC];
      vlc<RAG, T, VEC>(c, u_bot + c.i, t);
      const T left = (c.ix > 0) ? u_bot[c.i - 1] : T(0);
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const T l = (k > 0) ? t[(k + VEC - 1) % VEC] : left;
        kt[k] = t[k] * (-G.wx) + l * G.wx;
      }
      if (G.ndim >= 2) {
        const T *py = u_bot + G.n;
        vlc<RAG, T, VEC>(c, py + c.i, t);
        vzero(lo);
        if (c.iy > 0) vlc<RAG, T, VEC>(c, py + c.i - G.sy, lo);
#pragma unroll
        for (int k = 0; k < VEC; ++k) kt[k] += t[k] * (-G.wy) + lo[k] * G.wy;
      }
      if (G.ndim >= 3) {
        const T *pz = u_bot + 2 * G.n;
        vlc<RAG, T, VEC>(c, pz + c.i, t);
        vzero(lo);
        if (c.iz > 0) vlc<RAG, T, VEC>(c, pz + c.i - G.sz, lo);
#pragma unroll
        for (int k = 0; k < VEC; ++k) kt[k] += t[k] * (-G.wz) + lo[k] * G.wz;
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k) val[k] += c_btu * kt[k];
    }
    vlc<RAG, T, VEC>(c, v + c.i, t);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      val[k] += c_v * t[k];
      if (!RAG || k < c.nval) acc += (double)val[k] * (double)val[k];
    }
    vs<RAG, T, VEC>(c.nval, v_out + c.i, val);
    }
  }
  store_partial3(acc, ws);
}

// hbar = h + c_hbar*hbar ; x = x + c_x*hbar ; h = c_v*v + c_h*h ; sum x^2
// (4 bytes per lane and trip; 16 bytes only pay with one workgroup per CU: 0.64 ms)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_lsmr_hx(T *__restrict__ hbar,
                                                     T *__restrict__ x,
                                                     T *__restrict__ h,
                                                     const T *__restrict__ v,
                                                     int64_t n, T c_hbar, T c_x,
                                                     T c_h, T c_v, double *ws) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += stride) {
    const T hv = h[i];
    const T hb = c_hbar * hbar[i] + hv;
    hbar[i] = hb;
    const T xv = x[i] + c_x * hb;
    x[i] = xv;
    h[i] = c_h * hv + c_v * v[i];
    acc += (double)xv * (double)xv;
  }
  store_partial(acc, ws);
}

// grad = g + alpha * K^T(K x) and sum |K x|^2 in one pass over x (K = gradient
// with zero padding): the regulariser's share of cost and gradient of the
// robust-loss objective (tikhonov_linear_solver.py:201-208 with B = grad).
// Every difference is rounded as k_grad stores it and combined in the order of
// k_grad_adj and nsol_lincomb2, so grad equals the three-kernel result bit for bit.
template <typename T, int VEC, int ROWS, bool RAG, int MODE = 0>
__global__ __launch_bounds__(kBlock) void k_tk1_reg(const T *__restrict__ x,
                                                     const T *g, T *grad, Geom<T> G,
                                                     T alpha, double *ws,
                                                     const T *z = nullptr, T c_g = T(1),
                                                     T c_x = T(0), T c_z = T(0),
                                                     const T *gold = nullptr,
                                                     T *ydiff = nullptr) {
  // MODE 0: grad = g + alpha K'K x, sum |K x|^2.  MODE 1: the sum only (nothing read
  // but x, nothing written).  MODE 2: the three-term Lanczos update of the normal
  // equations, grad = c_g g + alpha K'K x + c_x x + c_z z (z may be null) with the
  // sum of squares of the RESULT -- one pass where MODE 0 and a combination took two.
  // MODE 3: MODE 0 with what L-BFGS-B asks of every new gradient while it is in
  // registers: its product with the search direction z (may be null) and the largest
  // |projected gradient| for the bounds c_x <= x <= c_z (+-inf: none), as
  // nsol_dot_* / nsol_lb_projgr_* form them, and (gold, ydiff given) the change of the
  // gradient ydiff = grad - gold with its sum of squares as nsol_lb_diff_dots_* does;
  // four partials per workgroup, kReducePartials doubles apart.
  const int64_t nrg = row_groups<T, VEC, ROWS>(G);
  double acc = 0.0, accd = 0.0, pg = 0.0, accy = 0.0;
  for (int64_t rg = blockIdx.y; rg < nrg; rg += gridDim.y) {
    const Voxel c = voxel_at<T, VEC, ROWS>(G, rg);
    if (!c.ok) continue;
    T v[VEC], nb[VEC], d[VEC], dp[VEC], out[VEC];
    vlc<RAG, T, VEC>(c, x + c.i, v);
    // x: forward differences at the lane's voxels and at the voxel to the left
    const T right = (c.ix + VEC < G.nx) ? x[c.i + VEC] : T(0);
    fwd_diff_x<T, VEC>(v, right, G.wx, d);
    const T dleft = (c.ix > 0) ? v[0] * G.wx + x[c.i - 1] * (-G.wx) : T(0);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const T l = (k > 0) ? d[(k + VEC - 1) % VEC] : dleft;
      out[k] = d[k] * (-G.wx) + l * G.wx;
      if (MODE != 2 && (!RAG || k < c.nval)) acc += (double)d[k] * (double)d[k];
    }
    if (G.ndim >= 2) {
      vzero(nb);
      if (c.iy + 1 < G.ny) vlc<RAG, T, VEC>(c, x + c.i + G.sy, nb);
      fwd_diff<T, VEC>(v, nb, G.wy, d);
      vzero(dp);
      if (c.iy > 0) {
        vlc<RAG, T, VEC>(c, x + c.i - G.sy, nb);
        fwd_diff<T, VEC>(nb, v, G.wy, dp);
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        out[k] += d[k] * (-G.wy) + dp[k] * G.wy;
        if (MODE != 2 && (!RAG || k < c.nval)) acc += (double)d[k] * (double)d[k];
      }
    }
    if (G.ndim >= 3) {
      vzero(nb);
      if (c.iz + 1 < G.nz) vlc<RAG, T, VEC>(c, x + c.i + G.sz, nb);
      fwd_diff<T, VEC>(v, nb, G.wz, d);
      vzero(dp);
      if (c.iz > 0) {
        vlc<RAG, T, VEC>(c, x + c.i - G.sz, nb);
        fwd_diff<T, VEC>(nb, v, G.wz, dp);
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        out[k] += d[k] * (-G.wz) + dp[k] * G.wz;
        if (MODE != 2 && (!RAG || k < c.nval)) acc += (double)d[k] * (double)d[k];
      }
    }
    if constexpr (MODE == 1) continue;
    vlc<RAG, T, VEC>(c, g + c.i, nb);
    if constexpr (MODE == 0 || MODE == 3) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) out[k] = T(1) * nb[k] + alpha * out[k];
      if constexpr (MODE == 3) {
        if (z) {
          vlc<RAG, T, VEC>(c, z + c.i, nb);
#pragma unroll
          for (int k = 0; k < VEC; ++k)
            if (!RAG || k < c.nval) accd += (double)out[k] * (double)nb[k];
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k)
          if (!RAG || k < c.nval) {
            T gi = out[k];
            if (gi < T(0)) {
              const T u = v[k] - c_z;
              gi = u > gi ? u : gi;
            } else {
              const T l = v[k] - c_x;
              gi = l < gi ? l : gi;
            }
            pg = fmax(pg, fabs((double)gi));
          }
        if (gold) {
          vlc<RAG, T, VEC>(c, gold + c.i, nb);
#pragma unroll
          for (int k = 0; k < VEC; ++k) {
            nb[k] = T(1) * out[k] + T(-1) * nb[k];
            if (!RAG || k < c.nval) accy += (double)nb[k] * (double)nb[k];
          }
          vs<RAG, T, VEC>(c.nval, ydiff + c.i, nb);
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < VEC; ++k) out[k] = c_g * nb[k] + alpha * out[k] + c_x * v[k];
      if (z) {
        vlc<RAG, T, VEC>(c, z + c.i, nb);
#pragma unroll
        for (int k = 0; k < VEC; ++k) out[k] += c_z * nb[k];
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k)
        if (!RAG || k < c.nval) acc += (double)out[k] * (double)out[k];
    }
    vs<RAG, T, VEC>(c.nval, grad + c.i, out);
  }
  if constexpr (MODE == 3) {
    // (one after the other: store_partial3's scratch is shared)
    store_partial3(acc, ws);
    __syncthreads();
    store_partial3(accd, ws + kReducePartials);
    __syncthreads();
    pg = wave_max_d(pg);
    __shared__ double smax[kBlock / kWave];
    if ((threadIdx.x & (kWave - 1)) == 0) smax[threadIdx.x / kWave] = pg;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
      for (int k = 0; k < kBlock / kWave; ++k) t = fmax(t, smax[k]);
      ws[2 * (int64_t)kReducePartials + (int64_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
    }
    __syncthreads();
    store_partial3(accy, ws + 3 * (int64_t)kReducePartials);
  } else {
    store_partial3(acc, ws);
  }
}

// result[0], [1], [3]: sums of the first, second and fourth run of partials; result[2]:
// maximum of the third (one workgroup each, fixed order)
__global__ __launch_bounds__(1024) void k_final_objective(const double *ws, int64_t nparts,
                                                           double *result) {
  const bool is_max = blockIdx.x == 2;
  const double *src = ws + (int64_t)blockIdx.x * kReducePartials;
  double v = 0.0;
  for (int64_t k = threadIdx.x; k < nparts; k += 1024) v = is_max ? fmax(v, src[k]) : v + src[k];
  __shared__ double s[16];
  v = is_max ? wave_max_d(v) : wave_sum_d(v);
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  if (lane == 0) s[wv] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < 16; ++k) t = is_max ? fmax(t, s[k]) : t + s[k];
    result[blockIdx.x] = t;
  }
}

inline int rgrid(int64_t n) {
  int g = grid_for(n);
  return g > kReducePartials ? kReducePartials : g;
}

// The stencil-mapped kernels leave one partial per workgroup; the caller's
// workspace holds kReducePartials doubles, so volumes with more workgroups
// than that reduce through a private scratch area past it is NOT available:
// instead rows are folded -- see ws_blocks().
template <int VEC, int ROWS>
inline int64_t grid_blocks(int64_t nz, int64_t ny, int64_t nx, int cap = 0) {
  const dim3 g = stencil_grid<VEC, ROWS>(nz, ny, nx, cap);
  return (int64_t)g.x * g.y * g.z;
}

template <typename T>
int u_impl(const T *Av, const T *v, T *u_top, T *u_bot, int bmode, int ndim,
           int64_t nz, int64_t ny, int64_t nx, double wx, double wy, double wz,
           double c_av, double c_bv, double c_u, double *result, double *ws,
           void *stream) {
  NSOL_CHECK_GEOM(ndim, nz, ny, nx);
  // Av == NULL: only the lower block is updated (and summed); needs one
  if (!u_top || !result || !ws || bmode < 0 || bmode > 2 ||
      (bmode != kBNone && (!v || !u_bot)) || (!Av && bmode == kBNone))
    return NSOL_EINVAL;
  const Geom<T> G = make_geom<T>(ndim, nz, ny, nx, wx, wy, wz);
  const bool al = (!Av || ptr16(Av)) && ptr16(u_top) && (!v || ptr16(v)) &&
                  (!u_bot || ptr16(u_bot)) && G.n % 4 == 0;
  return dispatch_stencil<T>(nz, ny, nx, al, [&](auto vec, auto rows, auto rag) {
    constexpr int V = decltype(vec)::value, R = decltype(rows)::value;
    constexpr bool RG = decltype(rag)::value;
    const int64_t nb = grid_blocks<V, R>(nz, ny, nx);
    hipLaunchKernelGGL((k_lsmr_u<T, V, R, RG>), (stencil_grid<V, R>(nz, ny, nx)),
                       dim3(kBlock), 0, as_stream(stream), Av, v, u_top, u_bot, G,
                       bmode, (T)c_av, (T)c_bv, (T)c_u, ws);
    hipLaunchKernelGGL(k_final_big, dim3(1), dim3(1024), 0, as_stream(stream), ws,
                       nb, result);
    return launch_status();
  });
}

template <typename T>
int v_impl(const T *Atu, const T *u_bot, const T *v, T *v_out, int bmode, int ndim,
           int64_t nz, int64_t ny, int64_t nx, double wx, double wy, double wz,
           double c_atu, double c_btu, double c_v, double *result, double *ws,
           void *stream) {
  NSOL_CHECK_GEOM(ndim, nz, ny, nx);
  if (!Atu || !v || !v_out || !result || !ws || bmode < 0 || bmode > 2 ||
      (bmode != kBNone && !u_bot))
    return NSOL_EINVAL;
  const Geom<T> G = make_geom<T>(ndim, nz, ny, nx, wx, wy, wz);
  const bool al = ptr16(Atu) && ptr16(v) && ptr16(v_out) && (!u_bot || ptr16(u_bot)) &&
                  G.n % 4 == 0;
  return dispatch_stencil<T>(nz, ny, nx, al, [&](auto vec, auto rows, auto rag) {
    constexpr int V = decltype(vec)::value, R = decltype(rows)::value;
    constexpr bool RG = decltype(rag)::value;
    const int64_t nb = grid_blocks<V, R>(nz, ny, nx);
    hipLaunchKernelGGL((k_lsmr_v<T, V, R, RG>), (stencil_grid<V, R>(nz, ny, nx)),
                       dim3(kBlock), 0, as_stream(stream), Atu, u_bot, v, v_out, G,
                       bmode, (T)c_atu, (T)c_btu, (T)c_v, ws);
    hipLaunchKernelGGL(k_final_big, dim3(1), dim3(1024), 0, as_stream(stream), ws,
                       nb, result);
    return launch_status();
  });
}

template <typename T>
int hx_impl(T *hbar, T *x, T *h, const T *v, int64_t n, double c_hbar,
            double c_x, double c_h, double c_v, double *result, double *ws,
            void *stream) {
  if (n < 1 || !hbar || !x || !h || !v || !result || !ws) return NSOL_EINVAL;
  // seven streams: fewer workgroups in flight keep the DRAM pages open longer
  // (512^3: 0.65 ms with 1024 workgroups, 0.70 with 2048, 0.73 with 4096)
  int g = rgrid(n);
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(k_lsmr_hx<T>, dim3(g), dim3(kBlock), 0, as_stream(stream),
                     hbar, x, h, v, n, (T)c_hbar, (T)c_x, (T)c_h, (T)c_v, ws);
  hipLaunchKernelGGL(k_final, dim3(1), dim3(kBlock), 0, as_stream(stream), ws, g,
                     result);
  return launch_status();
}

template <typename T>
int tk1_reg_impl(const T *x, const T *g, T *grad, int ndim, int64_t nz, int64_t ny,
                 int64_t nx, double wx, double wy, double wz, double alpha,
                 double *result, double *ws, void *stream) {
  NSOL_CHECK_GEOM(ndim, nz, ny, nx);
  if (!x || !g || !grad || !result || !ws || x == grad) return NSOL_EINVAL;
  const Geom<T> G = make_geom<T>(ndim, nz, ny, nx, wx, wy, wz);
  const bool al = ptr16(x) && ptr16(g) && ptr16(grad) && G.n % 4 == 0;
  return dispatch_stencil<T>(nz, ny, nx, al, [&](auto vec, auto rows, auto rag) {
    constexpr int V = decltype(vec)::value, R = decltype(rows)::value;
    constexpr bool RG = decltype(rag)::value;
    const int64_t nb = grid_blocks<V, R>(nz, ny, nx);
    hipLaunchKernelGGL((k_tk1_reg<T, V, R, RG>), (stencil_grid<V, R>(nz, ny, nx)),
                       dim3(kBlock), 0, as_stream(stream), x, g, grad, G, (T)alpha, ws);
    hipLaunchKernelGGL(k_final_big, dim3(1), dim3(1024), 0, as_stream(stream), ws,
                       nb, result);
    return launch_status();
  });
}

// MODE 3 of k_tk1_reg (see there): result[0] = sum |K x|^2, [1] = grad'd, [2] = max |proj grad|
template <typename T>
int tk1_objective_impl(const T *x, const T *g, T *grad, const T *d, const T *gold, T *ydiff,
                       int ndim, int64_t nz,
                       int64_t ny, int64_t nx, double wx, double wy, double wz, double alpha,
                       double lo, double hi, double *result, double *ws, void *stream) {
  NSOL_CHECK_GEOM(ndim, nz, ny, nx);
  if (!x || !g || !grad || !result || !ws || x == grad || d == grad || (gold && !ydiff) ||
      (gold && (gold == grad || ydiff == grad || ydiff == x || ydiff == g || ydiff == d)))
    return NSOL_EINVAL;
  const Geom<T> G = make_geom<T>(ndim, nz, ny, nx, wx, wy, wz);
  const bool al = ptr16(x) && ptr16(g) && ptr16(grad) && (!d || ptr16(d)) &&
                  (!gold || (ptr16(gold) && ptr16(ydiff))) && G.n % 4 == 0;
  return dispatch_stencil<T>(nz, ny, nx, al, [&](auto vec, auto rows, auto rag) {
    constexpr int V = decltype(vec)::value, R = decltype(rows)::value;
    constexpr bool RG = decltype(rag)::value;
    const int64_t nb = grid_blocks<V, R>(nz, ny, nx);
    hipLaunchKernelGGL((k_tk1_reg<T, V, R, RG, 3>), (stencil_grid<V, R>(nz, ny, nx)),
                       dim3(kBlock), 0, as_stream(stream), x, g, grad, G, (T)alpha, ws, d,
                       T(1), (T)lo, (T)hi, gold, ydiff);
    hipLaunchKernelGGL(k_final_objective, dim3(gold ? 4 : 3), dim3(1024), 0,
                       as_stream(stream), ws, nb, result);
    return launch_status();
  });
}

// MODE 1 / 2 of k_tk1_reg (see there)
template <typename T>
int tk1_norm_impl(const T *x, int ndim, int64_t nz, int64_t ny, int64_t nx, double wx,
                  double wy, double wz, double *result, double *ws, void *stream) {
  NSOL_CHECK_GEOM(ndim, nz, ny, nx);
  if (!x || !result || !ws) return NSOL_EINVAL;
  const Geom<T> G = make_geom<T>(ndim, nz, ny, nx, wx, wy, wz);
  const bool al = ptr16(x) && G.n % 4 == 0;
  return dispatch_stencil<T>(nz, ny, nx, al, [&](auto vec, auto rows, auto rag) {
    constexpr int V = decltype(vec)::value, R = decltype(rows)::value;
    constexpr bool RG = decltype(rag)::value;
    // (nothing is written but the partial sums: 16 384 workgroups, 0.16 against 0.19 ms)
    const int64_t nb = grid_blocks<V, R>(nz, ny, nx, 16384);
    hipLaunchKernelGGL((k_tk1_reg<T, V, R, RG, 1>), (stencil_grid<V, R>(nz, ny, nx, 16384)),
                       dim3(kBlock), 0, as_stream(stream), x, (const T *)nullptr,
                       (T *)nullptr, G, T(0), ws);
    hipLaunchKernelGGL(k_final_big, dim3(1), dim3(1024), 0, as_stream(stream), ws,
                       nb, result);
    return launch_status();
  });
}

template <typename T>
int tk1_lanczos_impl(const T *x, const T *g, const T *z, T *out, int ndim, int64_t nz,
                     int64_t ny, int64_t nx, double wx, double wy, double wz,
                     double alpha, double c_g, double c_x, double c_z, double *result,
                     double *ws, void *stream) {
  NSOL_CHECK_GEOM(ndim, nz, ny, nx);
  if (!x || !g || !out || !result || !ws || x == out || z == out) return NSOL_EINVAL;
  const Geom<T> G = make_geom<T>(ndim, nz, ny, nx, wx, wy, wz);
  const bool al = ptr16(x) && ptr16(g) && ptr16(out) && (!z || ptr16(z)) && G.n % 4 == 0;
  return dispatch_stencil<T>(nz, ny, nx, al, [&](auto vec, auto rows, auto rag) {
    constexpr int V = decltype(vec)::value, R = decltype(rows)::value;
    constexpr bool RG = decltype(rag)::value;
    const int64_t nb = grid_blocks<V, R>(nz, ny, nx);
    hipLaunchKernelGGL((k_tk1_reg<T, V, R, RG, 2>), (stencil_grid<V, R>(nz, ny, nx)),
                       dim3(kBlock), 0, as_stream(stream), x, g, out, G, (T)alpha, ws, z,
                       (T)c_g, (T)c_x, (T)c_z);
    hipLaunchKernelGGL(k_final_big, dim3(1), dim3(1024), 0, as_stream(stream), ws,
                       nb, result);
    return launch_status();
  });
}

}  // namespace

extern "C" {
#define NSOL_LSMR_DEF(T, SUF)                                                    \
  int nsol_lsmr_u_update_##SUF(const T *Av, const T *v, T *u_top, T *u_bot,      \
                               int bmode, int ndim, int64_t nz, int64_t ny,      \
                               int64_t nx, double wx, double wy, double wz,      \
                               double c_av, double c_bv, double c_u,             \
                               double *result, double *ws, void *s) {            \
    return u_impl<T>(Av, v, u_top, u_bot, bmode, ndim, nz, ny, nx, wx, wy, wz,   \
                     c_av, c_bv, c_u, result, ws, s);                            \
  }                                                                              \
  int nsol_lsmr_v_update_##SUF(const T *Atu, const T *u_bot, T *v, int bmode,    \
                               int ndim, int64_t nz, int64_t ny, int64_t nx,     \
                               double wx, double wy, double wz, double c_atu,    \
                               double c_btu, double c_v, double *result,         \
                               double *ws, void *s) {                            \
    return v_impl<T>(Atu, u_bot, v, v, bmode, ndim, nz, ny, nx, wx, wy, wz,      \
                     c_atu, c_btu, c_v, result, ws, s);                          \
  }                                                                              \
  int nsol_lsmr_v_update_to_##SUF(const T *Atu, const T *u_bot, const T *v,      \
                                  T *v_out, int bmode, int ndim, int64_t nz,     \
                                  int64_t ny, int64_t nx, double wx, double wy,  \
                                  double wz, double c_atu, double c_btu,         \
                                  double c_v, double *result, double *ws,        \
                                  void *s) {                                     \
    return v_impl<T>(Atu, u_bot, v, v_out, bmode, ndim, nz, ny, nx, wx, wy, wz,  \
                     c_atu, c_btu, c_v, result, ws, s);                          \
  }                                                                              \
  int nsol_lsmr_hx_update_##SUF(T *hbar, T *x, T *h, const T *v, int64_t n,      \
                                double c_hbar, double c_x, double c_h,           \
                                double c_v, double *result, double *ws,          \
                                void *s) {                                       \
    return hx_impl<T>(hbar, x, h, v, n, c_hbar, c_x, c_h, c_v, result, ws, s);   \
  }
NSOL_LSMR_DEF(float, f32)
NSOL_LSMR_DEF(double, f64)
int nsol_tk1_reg_cost_grad_f32(const float *x, const float *g, float *grad, int ndim,
                               int64_t nz, int64_t ny, int64_t nx, double wx,
                               double wy, double wz, double alpha, double *result,
                               double *ws, void *stream) {
  return tk1_reg_impl<float>(x, g, grad, ndim, nz, ny, nx, wx, wy, wz, alpha, result,
                             ws, stream);
}
int nsol_tk1_reg_cost_grad_f64(const double *x, const double *g, double *grad,
                               int ndim, int64_t nz, int64_t ny, int64_t nx,
                               double wx, double wy, double wz, double alpha,
                               double *result, double *ws, void *stream) {
  return tk1_reg_impl<double>(x, g, grad, ndim, nz, ny, nx, wx, wy, wz, alpha, result,
                              ws, stream);
}
#define NSOL_TK1_DEF(T, SUF)                                                     \
  int nsol_tk1_grad_norm_##SUF(const T *x, int ndim, int64_t nz, int64_t ny,     \
                               int64_t nx, double wx, double wy, double wz,      \
                               double *result, double *ws, void *stream) {       \
    return tk1_norm_impl<T>(x, ndim, nz, ny, nx, wx, wy, wz, result, ws, stream); \
  }                                                                              \
  int nsol_tk1_lanczos_##SUF(const T *x, const T *g, const T *z, T *out,         \
                             int ndim, int64_t nz, int64_t ny, int64_t nx,       \
                             double wx, double wy, double wz, double alpha,      \
                             double c_g, double c_x, double c_z, double *result, \
                             double *ws, void *stream) {                         \
    return tk1_lanczos_impl<T>(x, g, z, out, ndim, nz, ny, nx, wx, wy, wz,       \
                               alpha, c_g, c_x, c_z, result, ws, stream);        \
  }                                                                              \
  int nsol_tk1_reg_objective_##SUF(const T *x, const T *g, T *grad, const T *d,  \
                                   const T *gold, T *ydiff,                      \
                                   int ndim, int64_t nz, int64_t ny, int64_t nx, \
                                   double wx, double wy, double wz, double alpha, \
                                   double lo, double hi, double *result,         \
                                   double *ws, void *stream) {                   \
    return tk1_objective_impl<T>(x, g, grad, d, gold, ydiff, ndim, nz, ny, nx,   \
                                 wx, wy, wz, alpha, lo, hi, result, ws, stream); \
  }
NSOL_TK1_DEF(float, f32)
NSOL_TK1_DEF(double, f64)
}
