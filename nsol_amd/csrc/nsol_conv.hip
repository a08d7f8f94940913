// Correlation kernels behind the Gaussian blur A = A^T (separable 1-D passes)
// and arbitrary user kernels (dense taps), with scipy.ndimage's boundary modes.
// reference call sites: linear_operators.py:60-68, 82-86 (ndimage.convolve).
#include "nsol_common.hpp"

using namespace nsol;

namespace {

constexpr int kMaxTaps = 129;

template <typename T>
struct Taps {
  T w[kMaxTaps];
};

// index of the sample that position i (possibly outside [0,n)) refers to under
// `mode`; -1 means "zero" (constant mode).
__device__ __forceinline__ int64_t map_index(int64_t i, int64_t n, int mode) {
  if (i >= 0 && i < n) return i;
  switch (mode) {
    case NSOL_MODE_WRAP: {
      int64_t j = i % n;
      return j < 0 ? j + n : j;
    }
    case NSOL_MODE_NEAREST:
      return i < 0 ? 0 : n - 1;
    case NSOL_MODE_REFLECT: {
      const int64_t per = 2 * n;
      int64_t j = i % per;
      if (j < 0) j += per;
      return j < n ? j : per - 1 - j;
    }
    case NSOL_MODE_MIRROR: {
      if (n == 1) return 0;
      const int64_t per = 2 * n - 2;
      int64_t j = i % per;
      if (j < 0) j += per;
      return j < n ? j : per - j;
    }
    default:
      return -1;
  }
}

// One output per thread; lanes run along x so every tap read is a coalesced row
// segment (axis 0/1) or a shifted copy of the same row (axis 2, L1-resident).
template <typename T>
__global__ __launch_bounds__(kBlock) void k_corr_axis(const T *__restrict__ x,
                                                       T *__restrict__ out,
                                                       int64_t nz, int64_t ny,
                                                       int64_t nx, int axis,
                                                       Taps<T> taps, int ntaps,
                                                       int centre, int mode) {
  const int64_t n = nz * ny * nx;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t len = axis == 2 ? nx : (axis == 1 ? ny : nz);
  const int64_t step = axis == 2 ? 1 : (axis == 1 ? nx : ny * nx);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += stride) {
    const int64_t ix = i % nx;
    const int64_t r = i / nx;
    const int64_t pos = axis == 2 ? ix : (axis == 1 ? r % ny : r / ny);
    const int64_t base = i - pos * step;
    T acc = T(0);
    const int64_t lo = pos - centre;
    if (lo >= 0 && lo + ntaps <= len) {  // interior: no index mapping
      const T *src = x + base + lo * step;
      for (int t = 0; t < ntaps; ++t) acc += taps.w[t] * src[t * step];
    } else {
      for (int t = 0; t < ntaps; ++t) {
        const int64_t j = map_index(lo + t, len, mode);
        if (j >= 0) acc += taps.w[t] * x[base + j * step];
      }
    }
    out[i] = acc;
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_corr_dense(
    const T *__restrict__ x, T *__restrict__ out, int64_t nz, int64_t ny,
    int64_t nx, const T *__restrict__ taps, int kz, int ky, int kx, int cz,
    int cy, int cx, int mode) {
  const int64_t n = nz * ny * nx;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += stride) {
    const int64_t ix = i % nx;
    const int64_t r = i / nx;
    const int64_t iy = r % ny;
    const int64_t iz = r / ny;
    T acc = T(0);
    for (int a = 0; a < kz; ++a) {
      const int64_t jz = map_index(iz + a - cz, nz, mode);
      if (jz < 0) continue;
      for (int b = 0; b < ky; ++b) {
        const int64_t jy = map_index(iy + b - cy, ny, mode);
        if (jy < 0) continue;
        const T *row = x + (jz * ny + jy) * nx;
        const T *tw = taps + ((int64_t)a * ky + b) * kx;
        for (int c = 0; c < kx; ++c) {
          const int64_t jx = map_index(ix + c - cx, nx, mode);
          if (jx >= 0) acc += tw[c] * row[jx];
        }
      }
    }
    out[i] = acc;
  }
}

template <typename T>
int corr_axis_impl(const T *x, T *out, int axis, int64_t nz, int64_t ny,
                   int64_t nx, const double *taps_host, int ntaps, int centre,
                   int mode, void *stream) {
  if (!x || !out || x == out || !taps_host || axis < 0 || axis > 2 || nz < 1 ||
      ny < 1 || nx < 1 || ntaps < 1 || ntaps > kMaxTaps || centre < 0 ||
      centre >= ntaps || mode < 0 || mode > 4)
    return NSOL_EINVAL;
  Taps<T> taps;
  for (int t = 0; t < kMaxTaps; ++t) taps.w[t] = t < ntaps ? (T)taps_host[t] : T(0);
  const int64_t n = nz * ny * nx;
  hipLaunchKernelGGL(k_corr_axis<T>, dim3(grid_for(n)), dim3(kBlock), 0,
                     as_stream(stream), x, out, nz, ny, nx, axis, taps, ntaps,
                     centre, mode);
  return launch_status();
}

template <typename T>
int corr_dense_impl(const T *x, T *out, int64_t nz, int64_t ny, int64_t nx,
                    const T *taps, int kz, int ky, int kx, int cz, int cy,
                    int cx, int mode, void *stream) {
  if (!x || !out || x == out || !taps || nz < 1 || ny < 1 || nx < 1 || kz < 1 ||
      ky < 1 || kx < 1 || cz < 0 || cz >= kz || cy < 0 || cy >= ky || cx < 0 ||
      cx >= kx || mode < 0 || mode > 4)
    return NSOL_EINVAL;
  const int64_t n = nz * ny * nx;
  hipLaunchKernelGGL(k_corr_dense<T>, dim3(grid_for(n)), dim3(kBlock), 0,
                     as_stream(stream), x, out, nz, ny, nx, taps, kz, ky, kx, cz,
                     cy, cx, mode);
  return launch_status();
}

}  // namespace

extern "C" {
int nsol_corr_axis_f32(const float *x, float *out, int axis, int64_t nz,
                       int64_t ny, int64_t nx, const double *taps_host,
                       int ntaps, int centre, int mode, void *stream) {
  return corr_axis_impl<float>(x, out, axis, nz, ny, nx, taps_host, ntaps,
                               centre, mode, stream);
}
int nsol_corr_axis_f64(const double *x, double *out, int axis, int64_t nz,
                       int64_t ny, int64_t nx, const double *taps_host,
                       int ntaps, int centre, int mode, void *stream) {
  return corr_axis_impl<double>(x, out, axis, nz, ny, nx, taps_host, ntaps,
                                centre, mode, stream);
}
int nsol_corr_dense_f32(const float *x, float *out, int64_t nz, int64_t ny,
                        int64_t nx, const float *taps, int kz, int ky, int kx,
                        int cz, int cy, int cx, int mode, void *stream) {
  return corr_dense_impl<float>(x, out, nz, ny, nx, taps, kz, ky, kx, cz, cy, cx,
                                mode, stream);
}
int nsol_corr_dense_f64(const double *x, double *out, int64_t nz, int64_t ny,
                        int64_t nx, const double *taps, int kz, int ky, int kx,
                        int cz, int cy, int cx, int mode, void *stream) {
  return corr_dense_impl<double>(x, out, nz, ny, nx, taps, kz, ky, kx, cz, cy,
                                 cx, mode, stream);
}
}
