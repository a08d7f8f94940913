// Fused Golub-Kahan / LSMR vector kernels for the augmented Tikhonov system
//   [A; sqrt(alpha) B] x = [b; sqrt(alpha) b_reg]
// (reference: tikhonov_linear_solver.py:226-274 + scipy lsmr.py:320-413).
// Each kernel makes ONE pass, applies B = grad / B = identity on the fly and
// leaves the squared norm of its output as deterministic fp64 partial sums, so
// one LSMR iteration costs 3 launches + the blur passes of A and A^T instead
// of ~14 axpy/scale/dot launches.  The bidiagonalisation vectors are kept
// UNNORMALISED in memory (u~ = beta*u, v~ = alpha*v); the 1/beta, 1/alpha
// factors ride in the coefficients of the consuming kernel.
#include "nsol_common.hpp"

using namespace nsol;

namespace {

constexpr int kBNone = 0, kBGrad = 1, kBIdentity = 2;

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
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

// u_top = c_av*Av + c_u*u_top ; u_bot = c_bv*B(v) + c_u*u_bot ; sum of squares
template <typename T>
__global__ __launch_bounds__(kBlock) void k_lsmr_u(
    const T *__restrict__ Av, const T *__restrict__ v, T *__restrict__ u_top,
    T *__restrict__ u_bot, Geom<T> G, int bmode, T c_av, T c_bv, T c_u,
    double *ws) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < G.n;
       i += stride) {
    const T ut = c_av * Av[i] + c_u * u_top[i];
    u_top[i] = ut;
    acc += (double)ut * (double)ut;
    if (bmode == kBIdentity) {
      const T ub = c_bv * v[i] + c_u * u_bot[i];
      u_bot[i] = ub;
      acc += (double)ub * (double)ub;
    } else if (bmode == kBGrad) {
      const int64_t ix = i % G.nx;
      const int64_t r = i / G.nx;
      const T c = v[i];
      {
        const T nb = (ix + 1 < G.nx) ? v[i + 1] : T(0);
        const T ub = c_bv * (nb * G.wx + c * (-G.wx)) + c_u * u_bot[i];
        u_bot[i] = ub;
        acc += (double)ub * (double)ub;
      }
      if (G.ndim >= 2) {
        const T nb = (r % G.ny + 1 < G.ny) ? v[i + G.sy] : T(0);
        const T ub = c_bv * (nb * G.wy + c * (-G.wy)) + c_u * u_bot[G.n + i];
        u_bot[G.n + i] = ub;
        acc += (double)ub * (double)ub;
      }
      if (G.ndim >= 3) {
        const T nb = (r / G.ny + 1 < G.nz) ? v[i + G.sz] : T(0);
        const T ub =
            c_bv * (nb * G.wz + c * (-G.wz)) + c_u * u_bot[2 * G.n + i];
        u_bot[2 * G.n + i] = ub;
        acc += (double)ub * (double)ub;
      }
    }
  }
  store_partial(acc, ws);
}

// v = c_atu*Atu + c_btu*B^T(u_bot) + c_v*v ; sum of squares
template <typename T>
__global__ __launch_bounds__(kBlock) void k_lsmr_v(
    const T *__restrict__ Atu, const T *__restrict__ u_bot, T *__restrict__ v,
    Geom<T> G, int bmode, T c_atu, T c_btu, T c_v, double *ws) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < G.n;
       i += stride) {
    T val = c_atu * Atu[i];
    if (bmode == kBIdentity) {
      val += c_btu * u_bot[i];
    } else if (bmode == kBGrad) {
      const int64_t ix = i % G.nx;
      const int64_t r = i / G.nx;
      T kt = u_bot[i] * (-G.wx) + ((ix > 0) ? u_bot[i - 1] : T(0)) * G.wx;
      if (G.ndim >= 2) {
        const T *py = u_bot + G.n;
        kt += py[i] * (-G.wy) + ((r % G.ny > 0) ? py[i - G.sy] : T(0)) * G.wy;
      }
      if (G.ndim >= 3) {
        const T *pz = u_bot + 2 * G.n;
        kt += pz[i] * (-G.wz) + ((r / G.ny > 0) ? pz[i - G.sz] : T(0)) * G.wz;
      }
      val += c_btu * kt;
    }
    val += c_v * v[i];
    v[i] = val;
    acc += (double)val * (double)val;
  }
  store_partial(acc, ws);
}

// hbar = h + c_hbar*hbar ; x = x + c_x*hbar ; h = c_v*v + c_h*h ; sum x^2
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

inline int rgrid(int64_t n) {
  int g = grid_for(n);
  return g > kReducePartials ? kReducePartials : g;
}

template <typename T>
int u_impl(const T *Av, const T *v, T *u_top, T *u_bot, int bmode, int ndim,
           int64_t nz, int64_t ny, int64_t nx, double wx, double wy, double wz,
           double c_av, double c_bv, double c_u, double *result, double *ws,
           void *stream) {
  NSOL_CHECK_GEOM(ndim, nz, ny, nx);
  if (!Av || !u_top || !result || !ws || bmode < 0 || bmode > 2 ||
      (bmode != kBNone && (!v || !u_bot)))
    return NSOL_EINVAL;
  const Geom<T> G = make_geom<T>(ndim, nz, ny, nx, wx, wy, wz);
  const int g = rgrid(G.n);
  hipLaunchKernelGGL(k_lsmr_u<T>, dim3(g), dim3(kBlock), 0, as_stream(stream),
                     Av, v, u_top, u_bot, G, bmode, (T)c_av, (T)c_bv, (T)c_u, ws);
  hipLaunchKernelGGL(k_final, dim3(1), dim3(kBlock), 0, as_stream(stream), ws, g,
                     result);
  return launch_status();
}

template <typename T>
int v_impl(const T *Atu, const T *u_bot, T *v, int bmode, int ndim, int64_t nz,
           int64_t ny, int64_t nx, double wx, double wy, double wz,
           double c_atu, double c_btu, double c_v, double *result, double *ws,
           void *stream) {
  NSOL_CHECK_GEOM(ndim, nz, ny, nx);
  if (!Atu || !v || !result || !ws || bmode < 0 || bmode > 2 ||
      (bmode != kBNone && !u_bot))
    return NSOL_EINVAL;
  const Geom<T> G = make_geom<T>(ndim, nz, ny, nx, wx, wy, wz);
  const int g = rgrid(G.n);
  hipLaunchKernelGGL(k_lsmr_v<T>, dim3(g), dim3(kBlock), 0, as_stream(stream),
                     Atu, u_bot, v, G, bmode, (T)c_atu, (T)c_btu, (T)c_v, ws);
  hipLaunchKernelGGL(k_final, dim3(1), dim3(kBlock), 0, as_stream(stream), ws, g,
                     result);
  return launch_status();
}

template <typename T>
int hx_impl(T *hbar, T *x, T *h, const T *v, int64_t n, double c_hbar,
            double c_x, double c_h, double c_v, double *result, double *ws,
            void *stream) {
  if (n < 1 || !hbar || !x || !h || !v || !result || !ws) return NSOL_EINVAL;
  const int g = rgrid(n);
  hipLaunchKernelGGL(k_lsmr_hx<T>, dim3(g), dim3(kBlock), 0, as_stream(stream),
                     hbar, x, h, v, n, (T)c_hbar, (T)c_x, (T)c_h, (T)c_v, ws);
  hipLaunchKernelGGL(k_final, dim3(1), dim3(kBlock), 0, as_stream(stream), ws, g,
                     result);
  return launch_status();
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
    return v_impl<T>(Atu, u_bot, v, bmode, ndim, nz, ny, nx, wx, wy, wz, c_atu,  \
                     c_btu, c_v, result, ws, s);                                 \
  }                                                                              \
  int nsol_lsmr_hx_update_##SUF(T *hbar, T *x, T *h, const T *v, int64_t n,      \
                                double c_hbar, double c_x, double c_h,           \
                                double c_v, double *result, double *ws,          \
                                void *s) {                                       \
    return hx_impl<T>(hbar, x, h, v, n, c_hbar, c_x, c_h, c_v, result, ws, s);   \
  }
NSOL_LSMR_DEF(float, f32)
NSOL_LSMR_DEF(double, f64)
}
