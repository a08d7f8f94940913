// Shared helpers for the gfx950 kernels of libnsol_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nsol_hip.h"

namespace nsol {

constexpr int kWave = 64;            // CDNA wavefront
constexpr int kBlock = 256;          // 4 waves, one per SIMD
// grid-stride cap: 256 CUs x 8.  Element-wise kernels with 2..4 streams are 5-7 %
// faster than with 16 workgroups per CU (fewer concurrent DRAM streams), with 4
// per CU the 4-byte accesses no longer cover the latency (tools/tune_grid.py).
constexpr int kMaxGridBlocks = 2048;
constexpr int kMaxGridBlocksLimit = 4096;
inline int g_max_grid_blocks = kMaxGridBlocks;   // (runtime knob "max_grid_blocks")
inline int g_stencil_slabs = 1;                  // (runtime knob "stencil_slabs")
inline int g_stencil_blocks = 65536;             // (runtime knob "stencil_blocks": workgroups
                                                 // of a stencil kernel's grid, <= kReducePartials)
constexpr int kReducePartials = 65536;

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : static_cast<int>(e);
}

inline int grid_for(int64_t n, int per_block = kBlock) {
  int64_t b = (n + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > g_max_grid_blocks) b = g_max_grid_blocks;
  return static_cast<int>(b);
}

// Volume extents + inverse spacings handed to kernels by value.
template <typename T>
struct Geom {
  int64_t nz, ny, nx;
  int64_t sy;   // = nx        (row stride)
  int64_t sz;   // = ny*nx     (plane stride)
  int64_t n;    // = nz*ny*nx  (component stride of a gradient field)
  T wx, wy, wz; // 1/h
  int ndim;
  int slabs;    // stencil kernels deal the rows of a plane to the XCDs in slabs
                // (voxel_at in nsol_stencil.hpp; runtime knob "stencil_slabs")
  int padded;   // rows are held at a pitch sy > nx (whole 16-byte vectors): the
                // elements behind a row's end are padding, free to be overwritten
};

template <typename T>
inline Geom<T> make_geom(int ndim, int64_t nz, int64_t ny, int64_t nx,
                         double wx, double wy, double wz) {
  Geom<T> g;
  g.nz = nz; g.ny = ny; g.nx = nx;
  g.sy = nx; g.sz = ny * nx; g.n = nz * ny * nx;
  g.wx = static_cast<T>(wx); g.wy = static_cast<T>(wy); g.wz = static_cast<T>(wz);
  g.ndim = ndim;
  g.slabs = g_stencil_slabs;
  g.padded = 0;
  return g;
}

// The same volume with its rows at a pitch >= nx (elements): strides and the component
// stride of a gradient field follow the pitch, the extents stay.  pitch <= 0: contiguous.
template <typename T>
inline Geom<T> make_geom_pitched(int ndim, int64_t nz, int64_t ny, int64_t nx, int64_t pitch,
                                 double wx, double wy, double wz) {
  Geom<T> g = make_geom<T>(ndim, nz, ny, nx, wx, wy, wz);
  if (pitch > nx) {
    g.sy = pitch; g.sz = ny * pitch; g.n = nz * ny * pitch;
    g.padded = 1;
  }
  return g;
}

inline bool geom_ok(int ndim, int64_t nz, int64_t ny, int64_t nx) {
  if (ndim < 1 || ndim > 3 || nz < 1 || ny < 1 || nx < 1) return false;
  if (ndim < 3 && nz != 1) return false;
  if (ndim < 2 && ny != 1) return false;
  return true;
}

// ---- scalar maths with the reference's operation order (no FMA: the library
// is built with -ffp-contract=off so fp64 results are bit-comparable to NumPy)
template <typename T> __device__ __forceinline__ T t_abs(T v) { return v < T(0) ? -v : v; }
template <> __device__ __forceinline__ float t_abs<float>(float v) { return fabsf(v); }
template <> __device__ __forceinline__ double t_abs<double>(double v) { return fabs(v); }
template <typename T> __device__ __forceinline__ T t_max(T a, T b) { return a > b ? a : b; }
template <typename T> __device__ __forceinline__ T t_sqrt(T v);
template <> __device__ __forceinline__ float t_sqrt<float>(float v) { return sqrtf(v); }
template <> __device__ __forceinline__ double t_sqrt<double>(double v) { return sqrt(v); }

// x / max(1, |x|)   (proximal_operators.py:139-140), without the division:
// |x| <= 1 -> x / 1 = x exactly;  |x| > 1 -> x / |x| = +-1 exactly in IEEE
// arithmetic, so the result is bit-identical to the quotient.
// i.e. the median of (q, -1, 1): one v_med3_f32 in float.
template <typename T> __device__ __forceinline__ T dual_clamp(T q);
template <> __device__ __forceinline__ float dual_clamp<float>(float q) {
  return __builtin_amdgcn_fmed3f(q, -1.0f, 1.0f);
}
template <> __device__ __forceinline__ double dual_clamp<double>(double q) {
  return fmin(fmax(q, -1.0), 1.0);
}

// q / (1 + sigma*gamma)  (proximal_operators.py:157).  `hd` comes from
// huber_den<T>(1 + sigma*gamma): float64 divides as NumPy does (bit-comparable);
// float32 multiplies by the reciprocal formed in double on the host (<= 1 ulp from
// the quotient; an IEEE float division costs ~10 VALU instructions and the Huber
// kernels do one per dual component).  Every kernel uses these two helpers, so the
// fused, two-pass and generic forms stay bit-identical to each other.
template <typename T> inline T huber_den(double den);
template <> inline double huber_den<double>(double den) { return den; }
template <> inline float huber_den<float>(double den) { return (float)(1.0 / den); }
template <typename T> __device__ __forceinline__ T huber_div(T q, T hd);
template <> __device__ __forceinline__ double huber_div<double>(double q, double hd) { return q / hd; }
template <> __device__ __forceinline__ float huber_div<float>(float q, float hd) { return q * hd; }

// np.sign
template <typename T> __device__ __forceinline__ T t_sign(T v) {
  return v > T(0) ? T(1) : (v < T(0) ? T(-1) : T(0));
}

// prox of the data term (proximal_operators.py:95-98, 117-120)
template <typename T> __device__ __forceinline__ T prox_ell1(T u, T bt, T tl) {
  const T d = u - bt;
  return bt + t_max(t_abs(d) - tl, T(0)) * t_sign(d);
}
// (u + tl*bt) / (1 + tl)  (proximal_operators.py:117-120).  `den` comes from
// prox_den<T>(tl): float64 divides by 1 + tl as NumPy does (bit-comparable);
// float32 multiplies by the reciprocal formed in double on the host (<= 1 ulp
// from the quotient, ~9 fewer VALU instructions per voxel).
template <typename T> inline T prox_den(double tl);
template <> inline double prox_den<double>(double tl) { return 1.0 + tl; }
template <> inline float prox_den<float>(double tl) { return (float)(1.0 / (1.0 + tl)); }
template <typename T> __device__ __forceinline__ T prox_ell2(T u, T bt, T tl, T den);
template <> __device__ __forceinline__ double prox_ell2<double>(double u, double bt,
                                                                double tl, double den) {
  return (u + tl * bt) / den;
}
template <> __device__ __forceinline__ float prox_ell2<float>(float u, float bt,
                                                              float tl, float den) {
  return (u + tl * bt) * den;
}

// keeps `v` opaque so that the compiler cannot speculate the (expensive) code
// of a wave-uniform branch and select afterwards
template <typename T> __device__ __forceinline__ void pin(T &v) {
  asm volatile("" : "+v"(v));
}

template <typename T>
__device__ __forceinline__ T prox_data(T u, T bt, T tl, T one_plus_tl, bool l1) {
  if (l1) {
    pin(u);
    return prox_ell1(u, bt, tl);
  }
  pin(u);
  return prox_ell2(u, bt, tl, one_plus_tl);
}

}  // namespace nsol

#define NSOL_CHECK_GEOM(ndim, nz, ny, nx) \
  if (!nsol::geom_ok(ndim, nz, ny, nx)) return NSOL_EINVAL
