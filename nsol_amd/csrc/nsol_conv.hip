// Correlation kernels behind the Gaussian blur A = A^T (separable 1-D passes)
// and arbitrary user kernels (dense taps), with scipy.ndimage's boundary modes.
// reference call sites: linear_operators.py:60-68, 82-86 (ndimage.convolve).
#include <string.h>

#include <type_traits>

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


// ---------------------------------------------------------------------------
// Specialised periodic ("wrap") passes for odd tap counts with the centre in
// the middle -- every Gaussian of linear_operators.py:82-86.  16-byte
// accesses, taps unrolled at compile time, no integer division on the hot
// path; accumulation order t = 0..NT-1 as in ndimage.
//   * strided pass (array axis 0 or 1): a lane owns VEC consecutive x and RA
//     consecutive positions along the axis -> NT+RA-1 vector loads for RA
//     vector outputs (sliding window in registers);
//   * x pass (array axis 2): a lane owns VEC consecutive x and reads the
//     aligned vectors that cover [x-R, x+VEC-1+R]; neighbouring lanes read the
//     same lines, so all but ~1/(2*ceil(R/VEC)+1) of the loads are L1 hits.
// ---------------------------------------------------------------------------
template <typename T, int V>
struct VecOf {
  typedef T type __attribute__((ext_vector_type(V)));
};

__device__ __forceinline__ int64_t wrap_once(int64_t j, int64_t n) {
  // valid for -n <= j < 2n
  return j < 0 ? j + n : (j >= n ? j - n : j);
}

template <typename T, int VEC, int NT, int RA>
__global__ __launch_bounds__(kBlock) void k_corr_strided_wrap(
    const T *__restrict__ x, T *__restrict__ out, int64_t nz, int64_t ny,
    int64_t nx, int axis, Taps<T> taps) {
  typedef typename VecOf<T, VEC>::type V;
  constexpr int R = NT / 2;
  const int64_t nxv = nx / VEC;
  const int64_t xv = (int64_t)blockIdx.y * 64 + (threadIdx.x & 63);
  const int sub = threadIdx.x >> 6;   // 4 waves: 4 different lines
  // the two non-axis coordinates of this lane
  const int64_t len = axis == 0 ? nz : ny;
  const int64_t other = axis == 0 ? ny : nz;        // the remaining slow axis
  const int64_t o = (int64_t)blockIdx.x * 4 + sub;  // index along `other`
  const int64_t a0 = (int64_t)blockIdx.z * RA;      // first output along axis
  if (xv >= nxv || o >= other || a0 >= len) return;
  const int64_t step = axis == 0 ? ny * nx : nx;
  const int64_t ostep = axis == 0 ? nx : ny * nx;
  const T *base = x + o * ostep + xv * VEC;
  V win[NT + RA - 1];
#pragma unroll
  for (int t = 0; t < NT + RA - 1; ++t) {
    int64_t j = a0 - R + t;
    if (j < 0 || j >= len) { j %= len; if (j < 0) j += len; }
    win[t] = *reinterpret_cast<const V *>(base + j * step);
  }
#pragma unroll
  for (int r = 0; r < RA; ++r) {
    if (a0 + r < len) {
      V acc = taps.w[0] * win[r];
#pragma unroll
      for (int t = 1; t < NT; ++t) acc += taps.w[t] * win[r + t];
      *reinterpret_cast<V *>(out + o * ostep + xv * VEC + (a0 + r) * step) = acc;
    }
  }
}

template <typename T, int VEC, int NT, int XV>
__global__ __launch_bounds__(kBlock) void k_corr_x_wrap(
    const T *__restrict__ x, T *__restrict__ out, int64_t nrows, int64_t nx,
    Taps<T> taps) {
  // a lane produces XV adjacent output vectors from XV + 2*NBH input vectors
  typedef typename VecOf<T, VEC>::type V;
  constexpr int R = NT / 2;
  constexpr int NBH = (R + VEC - 1) / VEC;  // vectors on each side
  constexpr int NB = 2 * NBH + XV;
  const int64_t nxv = nx / VEC;
  const int64_t xv = ((int64_t)blockIdx.y * 64 + (threadIdx.x & 63)) * XV;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (xv >= nxv || row >= nrows) return;
  const T *rp = x + row * nx;
  T win[NB * VEC];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    int64_t j = xv + b - NBH;
    if (j < 0 || j >= nxv) { j %= nxv; if (j < 0) j += nxv; }
    const V v = *reinterpret_cast<const V *>(rp + j * VEC);
#pragma unroll
    for (int k = 0; k < VEC; ++k) win[b * VEC + k] = v[k];
  }
#pragma unroll
  for (int o = 0; o < XV; ++o) {
    if (xv + o < nxv) {
      V res;
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        T acc = taps.w[0] * win[(NBH + o) * VEC + k - R];
#pragma unroll
        for (int t = 1; t < NT; ++t)
          acc += taps.w[t] * win[(NBH + o) * VEC + k - R + t];
        res[k] = acc;
      }
      *reinterpret_cast<V *>(out + row * nx + (xv + o) * VEC) = res;
    }
  }
}

int g_blur3_zchunk = 0; // planes per z chunk of the one-pass blur; 0 = by the round model
int g_blur3_lxb = 16;   // lanes per row of the one-pass blur's tile (experiment knob)
int g_blur3_nw = 16;    // waves per workgroup of the LDS-DMA staged kernel (16 or 8)
int g_blur3_dma = 1;    // 1: LDS-DMA staged kernel for 16-byte rows; 0: k_blur3_wrap(_pp)
int g_corr_ra = 8;   // outputs per lane along a strided axis (experiment knob)
int g_corr_xv = 1;   // output vectors per lane in the x pass (experiment knob)

template <typename T, int VEC, int NT, int RA>
int launch_strided(const T *x, T *out, int axis, int64_t nz, int64_t ny,
                   int64_t nx, const Taps<T> &taps, hipStream_t st) {
  const int64_t nxv = nx / VEC;
  const int64_t len = axis == 0 ? nz : ny;
  const int64_t other = axis == 0 ? ny : nz;
  dim3 grid((unsigned)((other + 3) / 4), (unsigned)((nxv + 63) / 64),
            (unsigned)((len + RA - 1) / RA));
  hipLaunchKernelGGL((k_corr_strided_wrap<T, VEC, NT, RA>), grid, dim3(kBlock), 0,
                     st, x, out, nz, ny, nx, axis, taps);
  return launch_status();
}

template <typename T, int VEC, int NT>
int launch_wrap_nt(const T *x, T *out, int axis, int64_t nz, int64_t ny,
                   int64_t nx, const Taps<T> &taps, hipStream_t st) {
  if (axis != 2 && g_corr_ra == 8)
    return launch_strided<T, VEC, NT, 8>(x, out, axis, nz, ny, nx, taps, st);
  if (axis != 2 && g_corr_ra == 2)
    return launch_strided<T, VEC, NT, 2>(x, out, axis, nz, ny, nx, taps, st);
  if (axis == 2 && g_corr_xv == 2) {
    const int64_t nrows2 = nz * ny, nxv2 = nx / VEC;
    dim3 grid((unsigned)((nrows2 + 3) / 4), (unsigned)((nxv2 + 127) / 128), 1);
    hipLaunchKernelGGL((k_corr_x_wrap<T, VEC, NT, 2>), grid, dim3(kBlock), 0, st,
                       x, out, nrows2, nx, taps);
    return launch_status();
  }
  constexpr int RA = 4;
  const int64_t nxv = nx / VEC;
  if (axis == 2) {
    const int64_t nrows = nz * ny;
    dim3 grid((unsigned)((nrows + 3) / 4), (unsigned)((nxv + 63) / 64), 1);
    hipLaunchKernelGGL((k_corr_x_wrap<T, VEC, NT, 1>), grid, dim3(kBlock), 0, st,
                       x, out, nrows, nx, taps);
  } else {
    const int64_t len = axis == 0 ? nz : ny;
    const int64_t other = axis == 0 ? ny : nz;
    dim3 grid((unsigned)((other + 3) / 4), (unsigned)((nxv + 63) / 64),
              (unsigned)((len + RA - 1) / RA));
    hipLaunchKernelGGL((k_corr_strided_wrap<T, VEC, NT, RA>), grid, dim3(kBlock),
                       0, st, x, out, nz, ny, nx, axis, taps);
  }
  return launch_status();
}

// returns -2 when no specialisation applies
template <typename T>
int try_launch_wrap(const T *x, T *out, int axis, int64_t nz, int64_t ny,
                    int64_t nx, const Taps<T> &taps, int ntaps, int centre,
                    int mode, hipStream_t st) {
  constexpr int VEC = 16 / sizeof(T);
  if (mode != NSOL_MODE_WRAP || (ntaps & 1) == 0 || centre != ntaps / 2 ||
      nx % VEC != 0 || (reinterpret_cast<uintptr_t>(x) & 15u) ||
      (reinterpret_cast<uintptr_t>(out) & 15u))
    return -2;
  // grid limits: x < 2^31 blocks, y and z <= 65535
  if (nx / VEC > (int64_t)65535 * 64 || nz * ny > (int64_t)0x7fffffff) return -2;
  if (axis != 2 && (axis == 0 ? nz : ny) > (int64_t)65535 * 4) return -2;
#define NSOL_NT_CASE(N) \
  case N: return launch_wrap_nt<T, VEC, N>(x, out, axis, nz, ny, nx, taps, st);
  switch (ntaps) {
    NSOL_NT_CASE(3) NSOL_NT_CASE(5) NSOL_NT_CASE(7) NSOL_NT_CASE(9)
    NSOL_NT_CASE(11) NSOL_NT_CASE(13) NSOL_NT_CASE(15) NSOL_NT_CASE(17)
    NSOL_NT_CASE(19) NSOL_NT_CASE(21) NSOL_NT_CASE(23) NSOL_NT_CASE(25)
    default: return -2;
  }
#undef NSOL_NT_CASE
}

// ---------------------------------------------------------------------------
// Separable periodic 3-D correlation in ONE pass over memory (the Gaussian blur
// A = A^T of linear_operators.py:82-86 on a volume): reads x once, writes the
// result once (8 B per voxel instead of 24 for three 1-D passes).
//
// A workgroup owns a tile of TY rows x LXB*VEC voxels and marches along z:
//   x pass  every lane filters its own row segment straight from global memory
//           (raw buffer loads with 32-bit offsets; the taps' neighbours are L1
//           hits, as in k_corr_x_wrap) -- for the tile's rows and, in a second
//           round on the first waves, for the 2R halo rows -- and writes the
//           filtered vectors to LDS (double buffered, one barrier);
//   y pass  a lane reads the NT vectors of its column from LDS;
//   z pass  the xy-filtered values of the last NT - 1 planes live in registers
//           (a shifting window); once it is full every new plane yields one
//           output plane R planes behind.
// What bounds it (512^3, 13 taps, one plane per step: 0.42-0.43 ms = 2.5 TB/s of the
// 8 B per voxel; k_blur3_wrap_pp below takes two planes per step, 0.39 ms; switches
// compiled in for the measurement): loads and stores alone take 0.31 ms, the
// arithmetic alone 0.17 ms, and the two do not overlap -- one 16-wave workgroup
// per CU (the z window is 48 registers per lane) runs its phases in lock step.
// Prefetching the next plane's x windows needs 32 more registers and spills;
// fused multiply-adds, 8-wave workgroups and a tile with the halo rows on their
// own waves were measured and are no faster.
// Order of the passes: x, y, z, each accumulating t = 0..NT-1 like ndimage.
// (The three-kernel path runs z, y, x; both are rank-1 evaluations of the same
// dense kernel and differ from it, and from each other, by rounding only.)
// ---------------------------------------------------------------------------
// Raw buffer addressing for the one-pass blur: a 32-bit byte offset per lane
// plus a scalar plane offset; a lane whose offset is kNoLane is out of range of
// every buffer, so its load returns 0 without touching memory.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr uint32_t kNoLane = 0xC0000000u;
constexpr uint64_t kBlur3MaxBytes = 0xC0000000ull;

// The x window of one lane: NB aligned vectors around its own one.  With 4-wide
// vectors and R % 4 == 2 only half of the two outermost vectors is used, and
// only that half is loaded.
template <typename T, int VEC, int NT>
struct XWindow {
  static constexpr int R = NT / 2;
  static constexpr int NBH = (R + VEC - 1) / VEC;
  static constexpr int NB = 2 * NBH + 1;
  static constexpr bool HALF = (VEC == 4) && (R % VEC == 2);
  T v[NB * VEC];
  __device__ __forceinline__ void load(rsrc_t r, const uint32_t (&xo)[NB], uint32_t yo,
                                       uint32_t so) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      typedef typename VecOf<T, VEC>::type V;
      if constexpr (HALF) {
        if (b == 0 || b == NB - 1) {
          const int h = b == 0 ? VEC / 2 : 0;
          // (bit_cast the whole pair: applied to one element of a vector it
          // reads element 0 whatever the index)
          typedef T Pair __attribute__((ext_vector_type(2)));
          const Pair t = __builtin_bit_cast(Pair, __builtin_amdgcn_raw_buffer_load_b64(
              r, xo[b] + yo + (uint32_t)(h * sizeof(T)), so, 0));
          v[b * VEC + h] = t[0];
          v[b * VEC + h + 1] = t[1];
          continue;
        }
      }
      const V t = __builtin_bit_cast(
          V, __builtin_amdgcn_raw_buffer_load_b128(r, xo[b] + yo, so, 0));
#pragma unroll
      for (int k = 0; k < VEC; ++k) v[b * VEC + k] = t[k];
    }
  }
  // the same window element by element with the periodic wrap applied per
  // element: lanes at the ends of a row whose length is not a multiple of VEC
  // (ix0 = x index of v[0]; |ix0| < nx)
  __device__ __forceinline__ void load_edge(rsrc_t r, int ix0, int nx, uint32_t yo) {
#pragma unroll
    for (int i = NBH * VEC - R; i <= NBH * VEC + VEC - 1 + R; ++i) {
      int ix = ix0 + i;
      ix = ix < 0 ? ix + nx : (ix >= nx ? ix - nx : ix);
      if constexpr (sizeof(T) == 4)
        v[i] = __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(
                                         r, (uint32_t)ix * 4u + yo, 0, 0));
      else
        v[i] = __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b64(
                                         r, (uint32_t)ix * 8u + yo, 0, 0));
    }
  }
  __device__ __forceinline__ typename VecOf<T, VEC>::type filter(const Taps<T> &tx) const {
    typename VecOf<T, VEC>::type res;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      T acc = tx.w[0] * v[NBH * VEC + k - R];
#pragma unroll
      for (int t = 1; t < NT; ++t) acc += tx.w[t] * v[NBH * VEC + k - R + t];
      res[k] = acc;
    }
    return res;
  }
};

template <typename T, int VEC, int NT, int NW>
__global__ __launch_bounds__(NW * 64) void k_blur3_wrap(
    const T *__restrict__ x, T *__restrict__ out, int64_t nz, int64_t ny, int64_t nx,
    Taps<T> tz, Taps<T> ty, Taps<T> tx, int lxb, int ntx, int nty, int zchunk) {
  typedef typename VecOf<T, VEC>::type V;
  typedef XWindow<T, VEC, NT> W;
  constexpr int R = NT / 2;
  constexpr int NB = W::NB, NBH = W::NBH;
  constexpr int NT_THREADS = NW * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T *smem = reinterpret_cast<T *>(smem_raw);
  const int tyr = NT_THREADS / lxb;            // rows of the tile = rows of lanes
  const int frows = tyr + 2 * R;               // footprint rows
  const int rowlen = lxb * VEC;
  const int tid = threadIdx.x;
  const int row = tid / lxb;
  const int lx = tid - row * lxb;
  // (Plain tile order, x fastest: giving every XCD a contiguous run of tiles so
  // that shared halo lines meet in one L2 is slower here, 0.472 vs 0.456 ms.)
  int bid = blockIdx.x;
  const int bx = bid % ntx; bid /= ntx;
  const int by = bid % nty;
  const int bz = bid / nty;
  const int nxv = (int)(nx / VEC);
  const int xv = bx * lxb + lx;                        // own vector along x
  const int64_t y0 = (int64_t)by * tyr;
  const bool owner = xv < nxv && (y0 + row < ny);
  const uint32_t plane_bytes = (uint32_t)(ny * nx * sizeof(T));
  const int64_t plane = ny * nx;
  // one descriptor per plane (base = the plane, 32-bit offsets inside it): no
  // limit on the size of the volume
  auto plane_rsrc = [&](const T *base, int64_t z) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(base + z * plane), 0,
                                             plane_bytes, 0x00020000);
  };
  // wrapped byte offsets of the x window (loop invariant)
  uint32_t xo[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    int j = (xv + b - NBH) % nxv;
    if (j < 0) j += nxv;
    xo[b] = (uint32_t)j * (uint32_t)(VEC * sizeof(T));
  }
  // footprint rows this lane filters along x: `row` (round 0) and `tyr + row`
  // (round 1: the 2R halo rows, the first waves of the workgroup)
  const bool second = row < 2 * R;
  const bool second_wave = __builtin_amdgcn_readfirstlane((int)second) != 0;
  uint32_t yo[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    int64_t yy = (y0 - R + row + (int64_t)q * tyr) % ny;
    if (yy < 0) yy += ny;
    yo[q] = (uint32_t)(yy * nx * sizeof(T));
  }
  if (!second) yo[1] = kNoLane;                // lanes of a mixed wave: no access
  const int64_t zbeg = (int64_t)bz * zchunk;
  int64_t zend = zbeg + zchunk;
  if (zend > nz) zend = nz;
  int zw = (int)((zbeg - R) % nz);             // plane the x pass reads
  if (zw < 0) zw += (int)nz;
  V ring[NT - 1];                              // xy-filtered planes, oldest first
#pragma unroll
  for (int t = 0; t + 1 < NT; ++t) ring[t] = V(T(0));
  const int nsteps = (int)(zend - zbeg) + 2 * R;
  const uint32_t own_off =
      (uint32_t)((y0 + row) * nx * sizeof(T)) + (uint32_t)xv * (uint32_t)(VEC * sizeof(T));
  // (A window with compile-time slots -- the step loop unrolled NT times -- is no
  // faster at 13 taps and spills from 15 taps on.)
#pragma unroll 1
  for (int st = 0; st < nsteps; ++st) {
    T *buf = smem + (size_t)(st & 1) * frows * rowlen;
    const rsrc_t rs = plane_rsrc(x, zw);
    {
      W w;
      w.load(rs, xo, yo[0], 0);
      *reinterpret_cast<V *>(buf + (size_t)row * rowlen + lx * VEC) = w.filter(tx);
    }
    if (second_wave) {
      W w;
      w.load(rs, xo, yo[1], 0);
      const V r1 = w.filter(tx);
      if (second) *reinterpret_cast<V *>(buf + (size_t)(row + tyr) * rowlen + lx * VEC) = r1;
    }
    if (++zw == (int)nz) zw = 0;
    __syncthreads();
    const T *col = buf + (size_t)row * rowlen + lx * VEC;
    V v = ty.w[0] * *reinterpret_cast<const V *>(col);
#pragma unroll
    for (int t = 1; t < NT; ++t)
      v += ty.w[t] * *reinterpret_cast<const V *>(col + (size_t)t * rowlen);
    // z pass over the NT - 1 planes in the window and the new one; the window
    // then moves on by one plane
    if (st >= 2 * R && owner) {
      V acc = tz.w[0] * ring[0];
#pragma unroll
      for (int t = 1; t + 1 < NT; ++t) acc += tz.w[t] * ring[t];
      acc += tz.w[NT - 1] * v;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc),
                                             plane_rsrc(out, zbeg + (st - 2 * R)),
                                             own_off, 0, 0);
      asm volatile("s_nop 1");   // see nsol_pdk.hip: store data vs. the next VALU write
    }
#pragma unroll
    for (int t = 0; t + 2 < NT; ++t) ring[t] = ring[t + 1];
    ring[NT - 2] = v;
  }
}

// The same kernel with PP planes per step and barrier: the loads of both planes
// travel together and the z window moves by PP planes at a time (half the
// register moves per plane).  124 registers at 13 taps and PP = 2 -- used up to
// 13 taps (8-byte elements: 11); 0.39 instead of 0.43 ms at 512^3.
template <typename T, int VEC, int NT, int NW, int PP, bool RAGX>
__global__ __launch_bounds__(NW * 64) void k_blur3_wrap_pp(
    const T *__restrict__ x, T *__restrict__ out, int64_t nz, int64_t ny, int64_t nx,
    Taps<T> tz, Taps<T> ty, Taps<T> tx, int lxb, int ntx, int nty, int zchunk) {
  typedef typename VecOf<T, VEC>::type V;
  typedef XWindow<T, VEC, NT> W;
  constexpr int R = NT / 2;
  constexpr int NB = W::NB, NBH = W::NBH;
  constexpr int NT_THREADS = NW * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T *smem = reinterpret_cast<T *>(smem_raw);
  const int tyr = NT_THREADS / lxb;            // rows of the tile = rows of lanes
  const int frows = tyr + 2 * R;               // footprint rows
  const int rowlen = lxb * VEC;
  const int tid = threadIdx.x;
  const int row = tid / lxb;
  const int lx = tid - row * lxb;
  // (Plain tile order, x fastest: giving every XCD a contiguous run of tiles so
  // that shared halo lines meet in one L2 is slower here, 0.472 vs 0.456 ms.)
  int bid = blockIdx.x;
  const int bx = bid % ntx; bid /= ntx;
  const int by = bid % nty;
  const int bz = bid / nty;
  // RAGX: rows are not a multiple of VEC.  Vectors then start at 4-byte aligned
  // addresses (legal), the row's last vector holds nval < VEC valid elements,
  // and the lanes whose window crosses an end of the row load it element by
  // element with the wrap applied per element.
  const int nxv = RAGX ? (int)((nx + VEC - 1) / VEC) : (int)(nx / VEC);
  const int xv = bx * lxb + lx;                        // own vector along x
  const int64_t y0 = (int64_t)by * tyr;
  const bool owner = xv < nxv && (y0 + row < ny);
  int nval = VEC;
  bool edge = false;
  if constexpr (RAGX) {
    const int64_t left = nx - (int64_t)xv * VEC;
    nval = left >= VEC ? VEC : (left > 0 ? (int)left : 0);
    edge = xv < NBH || (int64_t)(xv + NBH + 1) * VEC > nx;
  }
  int ix0 = (xv - NBH) * VEC;                          // x index of the window's v[0]
  if constexpr (RAGX) {
    if (ix0 >= (int)nx) ix0 %= (int)nx;                // lanes past the row: any valid window
  }
  const uint32_t plane_bytes = (uint32_t)(ny * nx * sizeof(T));
  const int64_t plane = ny * nx;
  // one descriptor per plane (base = the plane, 32-bit offsets inside it): no
  // limit on the size of the volume
  auto plane_rsrc = [&](const T *base, int64_t z) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(base + z * plane), 0,
                                             plane_bytes, 0x00020000);
  };
  // wrapped byte offsets of the x window (loop invariant)
  uint32_t xo[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    int j = xv + b - NBH;
    if constexpr (!RAGX) {
      j %= nxv;
      if (j < 0) j += nxv;
    }
    xo[b] = (uint32_t)j * (uint32_t)(VEC * sizeof(T));   // (RAGX: edge lanes do not use it)
  }
  // footprint rows this lane filters along x: `row` (round 0) and `tyr + row`
  // (round 1: the 2R halo rows, the first waves of the workgroup)
  const bool second = row < 2 * R;
  const bool second_wave = __builtin_amdgcn_readfirstlane((int)second) != 0;
  uint32_t yo[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    int64_t yy = (y0 - R + row + (int64_t)q * tyr) % ny;
    if (yy < 0) yy += ny;
    yo[q] = (uint32_t)(yy * nx * sizeof(T));
  }
  if (!second) yo[1] = kNoLane;                // lanes of a mixed wave: no access
  const int64_t zbeg = (int64_t)bz * zchunk;
  int64_t zend = zbeg + zchunk;
  if (zend > nz) zend = nz;
  int zw = (int)((zbeg - R) % nz);             // plane the x pass reads
  if (zw < 0) zw += (int)nz;
  V ring[NT - 1];                              // xy-filtered planes, oldest first
#pragma unroll
  for (int t = 0; t + 1 < NT; ++t) ring[t] = V(T(0));
  const int nsteps = (int)(zend - zbeg) + 2 * R;
  const uint32_t own_off =
      (uint32_t)((y0 + row) * nx * sizeof(T)) + (uint32_t)xv * (uint32_t)(VEC * sizeof(T));
  // (A window with compile-time slots -- the step loop unrolled NT times -- is no
  // faster at 13 taps and spills from 15 taps on.)
  // PP planes per step (and per barrier): with two, their loads travel together
  // and the window moves by two planes at a time (half the register moves)
  const size_t tile = (size_t)frows * rowlen;            // one x-filtered plane
#pragma unroll 1
  for (int st = 0; st < nsteps; st += PP) {
    T *buf = smem + (size_t)((st / PP) & 1) * PP * tile;
    rsrc_t rs[PP];
#pragma unroll
    for (int q = 0; q < PP; ++q) {
      rs[q] = plane_rsrc(x, zw);
      if (++zw == (int)nz) zw = 0;     // (a step may read one valid plane too many)
    }
    // (opaque copy: sixteen loop-invariant edge offsets hoisted out of the loop
    // would all spill)
    int ix0v = ix0;
    if constexpr (RAGX) asm volatile("" : "+v"(ix0v));
    {
      W w[PP];
#pragma unroll
      for (int q = 0; q < PP; ++q) {
        if (RAGX && edge) w[q].load_edge(rs[q], ix0v, (int)nx, yo[0]);
        else w[q].load(rs[q], xo, yo[0], 0);
      }
#pragma unroll
      for (int q = 0; q < PP; ++q)
        *reinterpret_cast<V *>(buf + q * tile + (size_t)row * rowlen + lx * VEC) =
            w[q].filter(tx);
    }
    if (second_wave) {
      W w[PP];
#pragma unroll
      for (int q = 0; q < PP; ++q) {
        if (RAGX && edge) w[q].load_edge(rs[q], ix0v, (int)nx, yo[1]);
        else w[q].load(rs[q], xo, yo[1], 0);
      }
#pragma unroll
      for (int q = 0; q < PP; ++q) {
        const V r1 = w[q].filter(tx);
        if (second)
          *reinterpret_cast<V *>(buf + q * tile + (size_t)(row + tyr) * rowlen +
                                 lx * VEC) = r1;
      }
    }
    __syncthreads();
    V v[PP];
#pragma unroll
    for (int q = 0; q < PP; ++q) {
      const T *col = buf + q * tile + (size_t)row * rowlen + lx * VEC;
      v[q] = ty.w[0] * *reinterpret_cast<const V *>(col);
#pragma unroll
      for (int t = 1; t < NT; ++t)
        v[q] += ty.w[t] * *reinterpret_cast<const V *>(col + (size_t)t * rowlen);
    }
    // z pass: plane st + q sees the window's planes q.., then v[0..q]; the window
    // then moves on by PP planes
#pragma unroll
    for (int q = 0; q < PP; ++q) {
      const int sq = st + q;
      const int64_t z = zbeg + (sq - 2 * R);
      if (sq >= 2 * R && owner && (PP == 1 || z < zend)) {
        V acc = tz.w[0] * ring[q];
#pragma unroll
        for (int t = 1; t < NT; ++t) {
          if (t + q < NT - 1) acc += tz.w[t] * ring[t + q < NT - 1 ? t + q : 0];
          else acc += tz.w[t] * v[t + q - (NT - 1) < PP ? t + q - (NT - 1) : 0];
        }
        if (RAGX && nval < VEC) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const T ae = acc[e];   // (bit_cast of a vector element reads element 0)
            if (e < nval) {
              if constexpr (sizeof(T) == 4)
                __builtin_amdgcn_raw_buffer_store_b32(
                    __builtin_bit_cast(unsigned int, ae), plane_rsrc(out, z),
                    own_off + (uint32_t)(e * sizeof(T)), 0, 0);
              else
                __builtin_amdgcn_raw_buffer_store_b64(
                    __builtin_bit_cast(u32x2, ae), plane_rsrc(out, z),
                    own_off + (uint32_t)(e * sizeof(T)), 0, 0);
            }
          }
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc),
                                                 plane_rsrc(out, z), own_off, 0, 0);
        }
        asm volatile("s_nop 1");   // see nsol_pdk.hip: store data vs. the next VALU write
      }
    }
#pragma unroll
    for (int t = 0; t + PP < NT - 1; ++t) ring[t] = ring[t + PP];
#pragma unroll
    for (int q = 0; q < PP; ++q)
      if (NT - 1 - PP + q >= 0) ring[NT - 1 - PP + q] = v[q];
  }
}

// ---------------------------------------------------------------------------
// The one-pass blur with the input staged by LDS-DMA (k_blur3_dma).
//
// k_blur3_wrap_pp above is bound by two things its structure cannot fix: the x
// pass pulls five overlapping vectors per output vector through the L1, with the
// whole workgroup waiting out the HBM latency of every plane (no registers left
// for a prefetch: the z window holds 48), and its multiply / add pairs keep the
// SIMDs busy for 0.26 ms at 512^3 (276 vector instructions per wave and plane).
// Here
//   * the raw tile of plane s + 3 (rows and columns including the halo, periodic
//     wrap applied to the per-lane SOURCE address) travels global -> LDS with
//     global_load_lds_dwordx4 while plane s + 1 is filtered along x and plane s
//     along y and z: the loads cost no registers, are issued two phases before
//     their tile has to be complete (three raw tiles rotate; a counted
//     s_waitcnt vmcnt(N) leaves the newest one in flight across the barrier) and
//     every input byte crosses the L1 once;
//   * one barrier per plane: a phase runs the x pass of plane s + 1 (raw tile ->
//     x-filtered tile, both in LDS) and the y / z passes of plane s; the output of
//     plane s is stored at the START of the next phase, ahead of that phase's DMA;
//   * taps are applied with fused multiply-adds (v_pk_fma_f32 / v_fma_f64: half
//     the vector instructions; the blur is held to the reference by tolerance --
//     a separable evaluation of its dense kernel differs by rounding anyway);
//     (symmetric taps only -- every Gaussian; others take k_blur3_wrap_pp);
//   * tiles are dealt so that every XCD works on a run of consecutive tiles
//     (x fastest, then y): the halo columns and rows neighbouring tiles share are
//     then hits in that XCD's L2 instead of second trips to HBM.
// LDS per workgroup at 16 lanes per row, 13 taps, float: 3 raw tiles of 24 KiB +
// 2 x-filtered tiles of 19 KiB = 110 KiB.
// ---------------------------------------------------------------------------
template <typename V, typename T>
__device__ __forceinline__ V splat(T w) {
  V r;
#pragma unroll
  for (int k = 0; k < (int)(sizeof(V) / sizeof(T)); ++k) r[k] = w;
  return r;
}

__device__ __forceinline__ float fma1(float a, float b, float c) {
  return __builtin_fmaf(a, b, c);
}
__device__ __forceinline__ double fma1(double a, double b, double c) {
  return __builtin_fma(a, b, c);
}

constexpr int kDmaLxb = 16;

// phases U .. M-1 of one trip through the loop body (each with its position in the
// ring as a compile-time constant); stops at the end of the z chunk
template <int U, int M, typename F>
__device__ __forceinline__ void blur3_phases(int st0, int nsteps, F &f) {
  if constexpr (U < M) {
    if (st0 + U < nsteps) {
      f(st0 + U, std::integral_constant<int, U>());
      blur3_phases<U + 1, M>(st0, nsteps, f);
    }
  }
}

// ISO: the three axes share one set of taps (an isotropic Gaussian on unit or
// isotropic spacing -- BASELINE config 4): 14 fewer live scalars at 13 taps.
// EPI: instead of storing A x the kernel forms io = ca * (A x) + cb * io in place and
// the sum of squares of the result (per workgroup, in double: part[tile]) -- the top
// block of LSMR's u update, `u_top = c * A v + c' * u_top` and its norm
// (tikhonov_linear_solver.py:226-274 on SciPy's lsmr.py:320-336), without A v ever
// going to memory.  The old io tile of the next output plane is staged by LDS-DMA
// one phase ahead, issued BEFORE that phase's raw-tile pieces: the counted wait at
// the end of the phase leaves only younger operations in flight, so it has landed.
template <typename T, int VEC, int NT, int NW, bool ISO, bool EPI = false>
__global__ __launch_bounds__(NW * 64) void k_blur3_dma(
    const T *__restrict__ x, T *__restrict__ out, int64_t nz, int64_t ny, int64_t nx,
    Taps<T> tz_, Taps<T> ty_, Taps<T> tx, int ntx, int nty, int nzc, int zchunk,
    int per_xcd, T ca = T(1), T cb = T(0), double *__restrict__ part = nullptr) {
  const Taps<T> &tz = ISO ? tx : tz_;
  const Taps<T> &ty = ISO ? tx : ty_;
  typedef typename VecOf<T, VEC>::type V;
  constexpr int lxb = kDmaLxb;                 // lanes per tile row (compile time: the
                                               // LDS strides fold into the addresses)
  constexpr int R = NT / 2;
  constexpr int NBH = (R + VEC - 1) / VEC;     // halo vectors on each side of a row
  constexpr int NB = 2 * NBH + 1;
  constexpr int NTHR = NW * 64;
  constexpr int MAXP = 3;                      // LDS-DMA pieces per wave and plane
  // the taps are symmetric (checked on the host): tap t is read as w[min(t, NT-1-t)],
  // which leaves 3 * (R + 1) scalars live instead of 3 * NT (39 of them at 13 taps
  // overflow the scalar registers and come back as a v_readlane per use)
  auto sym = [](int t) { return t <= R ? t : NT - 1 - t; };
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // every XCD (block id mod 8) takes a run of per_xcd consecutive tiles
  const int total = ntx * nty * nzc;
  const int logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (logical >= total) return;
  const int bx = logical % ntx;
  const int by = (logical / ntx) % nty;
  const int bz = logical / (ntx * nty);

  constexpr int tyr = NTHR / lxb;              // rows of the tile = rows of lanes
  constexpr int frows = tyr + 2 * R;           // rows of the raw / x-filtered tile
  constexpr int rl = lxb + 2 * NBH;            // vectors per raw row
  constexpr int raw_vecs = frows * rl;
  constexpr int npieces = (raw_vecs + 63) >> 6;  // 1 KiB per wave-instruction
  constexpr int raw_stride = npieces * 64;     // vectors per raw buffer
  constexpr int xf_stride = frows * lxb;
  static_assert(npieces <= MAXP * NW, "raw tile needs more LDS-DMA pieces per wave");
  static_assert(2 * R <= tyr, "halo rows must fit one round of lanes");
  V *raw = reinterpret_cast<V *>(smem_raw);    // three raw tiles, then two x-filtered
  V *xf = raw + 3 * (size_t)raw_stride;
  constexpr int tile_vecs = tyr * lxb;         // (EPI) two tiles of the old io values
  V *obuf = xf + 2 * (size_t)xf_stride;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Lane -> (row, lx).  A wave covers 4 rows x 16 lanes, and a ds_read_b128 is
  // served in four groups of 16 lanes, {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and
  // the same + 32 (MI355X_MICROARCH.md, LDS): the map puts every group on ONE row,
  // so its 16 lanes read 16 consecutive 16-byte slots -- conflict-free whatever the
  // row stride (20 slots in the raw tile).  With the plain map (row = lane / 16) half
  // of each group sat a row further and a fifth of the LDS cycles were conflicts.
  const int lx = lane & 15;
  const int quad = (lane >> 2) & 3;
  const int rsel = ((quad == 1 || quad == 2) ? 1 : 0) ^ ((lane >> 4) & 1);
  const int row = (tid >> 6) * 4 + ((lane >> 5) & 1) * 2 + rsel;
  const int nxv = (int)(nx / VEC);
  const int xv = bx * lxb + lx;
  const int64_t y0 = (int64_t)by * tyr;
  const bool owner = xv < nxv && (y0 + row < ny);
  const int64_t plane = ny * nx;

  // LDS-DMA source offsets (elements inside a plane) of this lane's pieces:
  // piece k = wave + j * NW covers the raw vectors [64 k, 64 k + 64)
  uint32_t src_off[MAXP];
#pragma unroll
  for (int j = 0; j < MAXP; ++j) {
    // (lanes past the end of the raw tile re-load its first vector into the
    // padding behind it: no predicate to carry through the loop)
    int i = (wave + j * NW) * 64 + lane;
    if (i >= raw_vecs) i = 0;
    const int rr = i / rl;
    const int cc = i - rr * rl;
    int64_t yy = (y0 - R + rr) % ny;
    if (yy < 0) yy += ny;
    int xx = (bx * lxb - NBH + cc) % nxv;
    if (xx < 0) xx += nxv;
    src_off[j] = (uint32_t)(yy * nx + (int64_t)xx * VEC);
  }
  // pieces this wave issues per plane (wave-uniform)
  const int my_pieces = (npieces - wave + NW - 1) / NW;
  auto stage = [&](int64_t z, int rbuf) {      // plane z -> raw tile at vector offset rbuf
    const T *pl = x + z * plane;
#pragma unroll
    for (int j = 0; j < MAXP; ++j) {
      if (j * NW >= npieces) break;              // (compile time)
      const int k = wave + j * NW;
      if ((j + 1) * NW <= npieces || k < npieces)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void *)(pl + src_off[j]),
            (__attribute__((address_space(3))) void *)(raw + (size_t)rbuf + (size_t)k * 64),
            16, 0, 0);
    }
  };
  // End of a phase: the LDS writes of this phase are done (lgkmcnt) and at most
  // `newer` of this wave's vector-memory operations are still in flight.  On gfx9
  // vmcnt is decremented in issue order for loads and stores alike, so with
  // `newer` = the number of operations issued after the pieces of the tile that has
  // to be complete (<= MAXP pieces + 1 store), that tile has landed.  Then the barrier.
  auto phase_end = [&](int newer) {                // (wave-uniform)
    switch (newer) {
      case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    }
  };
  // x pass: raw tile -> x-filtered tile; a lane filters footprint row `row` and,
  // in the first waves, the halo row `tyr + row`
  const bool second = row < 2 * R;
  const bool second_wave = __builtin_amdgcn_readfirstlane((int)second) != 0;
  auto xrow = [&](const V *rb, V *xb, int fr) {
    const V *w = rb + (size_t)fr * rl + lx;
    constexpr int off = NBH * VEC - R;             // window index of output 0, tap 0
    V res;
    if constexpr (sizeof(T) == 4 && VEC == 4) {
      // Packed form: every vector instruction costs one issue slot whether it handles
      // one float or two, so the taps are applied to aligned PAIRS of the window.
      // Taps t with off + t even see outputs (0,1) and (2,3) on aligned pairs; the
      // others see them shifted by one element: they are summed on the pair grid
      // (B[0..2]) and their halves added to the outputs at the end.  32 + 4
      // instructions instead of 52 at 13 taps; the summation order differs from
      // t = 0 .. NT-1 (rounding only).
      typedef T P2 __attribute__((ext_vector_type(2)));
      P2 P[NB * 2];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const V t = w[b];
        P[2 * b] = P2{t[0], t[1]};
        P[2 * b + 1] = P2{t[2], t[3]};
      }
      P2 A[2], B[3];
      bool a_set = false, b_set = false;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const P2 wt = P2{tx.w[sym(t)], tx.w[sym(t)]};
        if (((off + t) & 1) == 0) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const P2 src = P[(off + 2 * h + t) / 2];
            A[h] = a_set ? __builtin_elementwise_fma(wt, src, A[h]) : wt * src;
          }
          a_set = true;
        } else {
          const int sidx = (off + t + 1) / 2;
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            const P2 src = P[sidx - 1 + a];
            B[a] = b_set ? __builtin_elementwise_fma(wt, src, B[a]) : wt * src;
          }
          b_set = true;
        }
      }
      if (!a_set) A[0] = A[1] = P2{T(0), T(0)};
      if (!b_set) B[0] = B[1] = B[2] = P2{T(0), T(0)};
      res[0] = A[0][0] + B[0][1];
      res[1] = A[0][1] + B[1][0];
      res[2] = A[1][0] + B[1][1];
      res[3] = A[1][1] + B[2][0];
    } else {
      T win[NB * VEC];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const V t = w[b];
#pragma unroll
        for (int k = 0; k < VEC; ++k) win[b * VEC + k] = t[k];
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        T acc = tx.w[0] * win[off + k];
#pragma unroll
        for (int t = 1; t < NT; ++t) acc = fma1(tx.w[sym(t)], win[off + k + t], acc);
        res[k] = acc;
      }
    }
    xb[(size_t)fr * lxb + lx] = res;
  };
  auto xpass = [&](int rbuf, int xbuf) {
    const V *rb = raw + (size_t)rbuf;
    V *xb = xf + (size_t)xbuf * xf_stride;
    xrow(rb, xb, row);
    if (second_wave) {
      if (second) xrow(rb, xb, row + tyr);
    }
  };

  const int64_t zbeg = (int64_t)bz * zchunk;
  int64_t zend = zbeg + zchunk;
  if (zend > nz) zend = nz;
  const int nsteps = (int)(zend - zbeg) + 2 * R;    // planes zbeg - R .. zend + R - 1
  int zw = (int)((zbeg - R) % nz);                  // plane of the next stage()
  if (zw < 0) zw += (int)nz;
  auto next_plane = [&]() {
    const int z = zw;
    if (++zw == (int)nz) zw = 0;
    return (int64_t)z;
  };
  V ring[NT - 1];                                   // xy-filtered planes, oldest first
#pragma unroll
  for (int t = 0; t + 1 < NT; ++t) ring[t] = splat<V, T>(T(0));
  // output: one buffer descriptor per plane and a 32-bit offset inside it; lanes
  // that own no voxel carry an out-of-range offset (the store is dropped by the
  // hardware), so EVERY wave issues exactly one store per output plane -- the
  // counted waits below depend on that
  const uint32_t plane_bytes = (uint32_t)(plane * sizeof(T));
  const uint32_t own_off =
      owner ? (uint32_t)(((y0 + row) * nx + (int64_t)xv * VEC) * sizeof(T)) : kNoLane;
  double sumsq = 0.0;
  auto put = [&](int64_t z, V val, int ob) {
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out + z * plane, 0, plane_bytes,
                                                        0x00020000);
    if constexpr (EPI) {
      const V old = obuf[(size_t)ob * tile_vecs + (size_t)row * lxb + lx];
      val = splat<V, T>(ca) * val + splat<V, T>(cb) * old;
      if (owner) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) sumsq += (double)val[e] * (double)val[e];
      }
    }
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), rs, own_off, 0, 0);
    asm volatile("s_nop 1");   // see nsol_pdk.hip: store data vs. the next VALU write
  };
  // (EPI) the io tile of one output plane -> obuf[ob]: one 1-KiB piece per wave; lanes
  // whose tile position lies outside the volume re-read a valid neighbour
  uint32_t old_off = 0;
  if constexpr (EPI) {
    const int i = wave * 64 + lane;
    int64_t yy = y0 + i / lxb;
    if (yy >= ny) yy = ny - 1;
    int xx = bx * lxb + i % lxb;
    if (xx >= nxv) xx = nxv - 1;
    old_off = (uint32_t)(yy * nx + (int64_t)xx * VEC);
  }
  auto stage_old = [&](int64_t z, int ob) {
    if constexpr (EPI)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void *)(out + z * plane + old_off),
          (__attribute__((address_space(3))) void *)(obuf + (size_t)ob * tile_vecs +
                                                     (size_t)wave * 64),
          16, 0, 0);
  };

  // prologue: planes 0, 1, 2 staged, plane 0 filtered along x  (nsteps >= 2R + 1 >= 3)
  stage(next_plane(), 0);
  stage(next_plane(), raw_stride);
  stage(next_plane(), 2 * raw_stride);
  phase_end(0);
  xpass(0, 0);
  phase_end(0);
  // rotating vector offsets of the raw tiles: r_cur holds plane st (free: the target
  // of this phase's DMA), r_next plane st + 1, r_after plane st + 2
  int r_cur = 0, r_next = raw_stride, r_after = 2 * raw_stride;
  // One phase = one plane and one barrier.  The loop body holds M = NT - 1 phases:
  // the z window is a ring of M register vectors whose slot indices are then
  // compile-time constants (no register moves: they were a third of the vector
  // instructions), like the x-filtered buffer's index.
  constexpr int M = NT - 1;
  auto phase = [&](int st, auto U) {
    constexpr int u = decltype(U)::value;           // = st mod M
    constexpr int q = u & 1;                        // = st & 1 (M is even)
    const bool more = st + 3 < nsteps;
    if (EPI && st + 1 >= 2 * R && st + 1 < nsteps)
      stage_old(zbeg + (st + 1 - 2 * R), q ^ 1);    // old io of the next output plane
    if (more) stage(next_plane(), r_cur);           // plane st + 3, two phases ahead
    // (The x pass of plane st + 1 and the y / z passes of plane st are independent,
    // but letting half of the waves of a SIMD run them in the opposite order, so that
    // not everybody waits for the LDS at the same time, measured no faster.)
    const bool storing = st >= 2 * R;
    auto yz = [&]() {
      const V *col = xf + (size_t)q * xf_stride + (size_t)row * lxb + lx;
      V v = splat<V, T>(ty.w[0]) * col[0];
#pragma unroll
      for (int t = 1; t < NT; ++t)
        v = __builtin_elementwise_fma(splat<V, T>(ty.w[sym(t)]), col[(size_t)t * lxb], v);
      if (st >= 2 * R) {
        // window of plane st: the ring from its oldest slot (u), then v
        V acc = splat<V, T>(tz.w[0]) * ring[u];
#pragma unroll
        for (int t = 1; t < M; ++t)
          acc = __builtin_elementwise_fma(splat<V, T>(tz.w[sym(t)]), ring[(u + t) % M], acc);
        acc = __builtin_elementwise_fma(splat<V, T>(tz.w[0]), v, acc);
        if (storing) put(zbeg + (st - 2 * R), acc, q);
      }
      ring[u] = v;                                  // replaces plane st - M
    };
    if (st + 1 < nsteps) xpass(r_next, q ^ 1);
    yz();
    const int t_ = r_cur; r_cur = r_next; r_next = r_after; r_after = t_;
    // plane st + 2 (staged in the previous phase) must have landed; younger than
    // its pieces are this phase's pieces and this phase's store
    phase_end((more ? my_pieces : 0) + (storing ? 1 : 0));
  };
#pragma unroll 1
  for (int st0 = 0; st0 < nsteps; st0 += M) blur3_phases<0, M>(st0, nsteps, phase);
  if constexpr (EPI) {
    // (the last phase ended with a barrier: the LDS is free)
    double *red = reinterpret_cast<double *>(smem_raw);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sumsq += __shfl_down(sumsq, o, 64);
    if (lane == 0) red[wave] = sumsq;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int w2 = 0; w2 < NW; ++w2) t += red[w2];
      part[logical] = t;
    }
  }
}

// sum of the per-tile partials in a fixed order
__global__ __launch_bounds__(kBlock) void k_blur3_epi_final(const double *part, int n,
                                                            double *result) {
  __shared__ double s[kBlock];
  double t = 0.0;
  for (int i = threadIdx.x; i < n; i += kBlock) t += part[i];
  s[threadIdx.x] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    double r = 0.0;
    for (int i = 0; i < kBlock; ++i) r += s[i];
    *result = r;
  }
}

inline int blur3_cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipGetDeviceProperties(&prop, dev) == hipSuccess)
      n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

// LDS-DMA staged kernel: tiles of kDmaLxb lanes per row whatever the row length;
// returns -2 when it does not apply.  EPI (out = io, in place): io = ca * blur(x) +
// cb * io and *result = sum of squares of the new io (part: >= tiles doubles).
template <typename T, int VEC, int NT, int NWD, bool EPI = false>
int launch_blur3_dma(const T *x, T *out, int64_t nz, int64_t ny, int64_t nx,
                     const Taps<T> &tz, const Taps<T> &ty, const Taps<T> &tx,
                     hipStream_t st, double ca = 1.0, double cb = 0.0,
                     double *result = nullptr, double *part = nullptr,
                     int64_t part_doubles = 0) {
  constexpr int R = NT / 2;
  constexpr int NBH = (R + VEC - 1) / VEC;
  constexpr int dl = kDmaLxb;
  constexpr int dtyr = (NWD * 64) / dl;
  constexpr int frows = dtyr + 2 * R;
  constexpr int npieces = (frows * (dl + 2 * NBH) + 63) / 64;
  constexpr size_t lds = (3 * (size_t)npieces * 64 + 2 * (size_t)frows * dl +
                          (EPI ? 2 * (size_t)dtyr * dl : 0)) * 16;
  if constexpr (lds > 160 * 1024) {
    static_assert(EPI, "LDS-DMA blur tile does not fit");
    return -2;                                           // (no room for the io tiles)
  } else {
  if (dtyr < 2 * R) return -2;
  const int64_t nxv = nx / VEC;
  const int64_t dntx = (nxv + dl - 1) / dl, dnty = (ny + dtyr - 1) / dtyr;
  if (ny * nx >= ((int64_t)1 << 31)) return -2;          // 32-bit offsets in a plane
  // z chunks by the round model: `slots` workgroups run at a time, a launch takes
  // ceil(workgroups / slots) rounds of (chunk + 2R) plane steps
  const int per_cu = (int)((160 * 1024) / lds) < (32 / NWD) ? (int)((160 * 1024) / lds)
                                                            : (32 / NWD);
  const int64_t slots = (int64_t)blur3_cu_count() * (per_cu < 1 ? 1 : per_cu);
  int64_t zchunk = nz, best = -1;
  for (int64_t c = 1; c <= nz && (nz + c - 1) / c >= R; ++c) {
    const int64_t len = (nz + c - 1) / c;
    const int64_t chunks = (nz + len - 1) / len;
    const int64_t rounds = (dntx * dnty * chunks + slots - 1) / slots;
    const int64_t cost = rounds * (len + 2 * R);
    if (best < 0 || cost < best) { best = cost; zchunk = len; }
    if (dntx * dnty * chunks >= 64 * slots) break;
  }
  if (g_blur3_zchunk > 0) zchunk = g_blur3_zchunk < nz ? g_blur3_zchunk : nz;
  const int64_t nzc = (nz + zchunk - 1) / zchunk;
  const int64_t tiles = dntx * dnty * nzc;
  if (tiles >= ((int64_t)1 << 28)) return -2;
  if (EPI && tiles > part_doubles) return -2;
  const int per_xcd = (int)((tiles + 7) / 8);
  bool iso = true;
  for (int t = 0; t < NT; ++t) iso = iso && tz.w[t] == tx.w[t] && ty.w[t] == tx.w[t];
  auto kern = iso ? k_blur3_dma<T, VEC, NT, NWD, true, EPI>
                  : k_blur3_dma<T, VEC, NT, NWD, false, EPI>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) { (void)hipGetLastError(); return -2; }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(per_xcd * 8)), dim3(NWD * 64), lds, st, x, out,
                     nz, ny, nx, tz, ty, tx, (int)dntx, (int)dnty, (int)nzc, (int)zchunk,
                     per_xcd, (T)ca, (T)cb, part);
  if (EPI)
    hipLaunchKernelGGL(k_blur3_epi_final, dim3(1), dim3(kBlock), 0, st, part, (int)tiles,
                       result);
  return launch_status();
  }
}

// nsol_corr3_wrap_axpby_*: io = ca * A x + cb * io in place with the sum of squares
// of the result; -2 when the LDS-DMA kernel does not apply (the caller then blurs
// and combines in two steps)
template <typename T>
int corr3_axpby_impl(const T *x, T *io, int64_t nz, int64_t ny, int64_t nx,
                     const double *tz_host, const double *ty_host, const double *tx_host,
                     int ntaps, double ca, double cb, double *result, double *ws,
                     int64_t ws_doubles, void *stream) {
  if (!x || !io || x == io || !tz_host || !ty_host || !tx_host || !result || !ws ||
      nz < 1 || ny < 1 || nx < 1 || ntaps < 1)
    return NSOL_EINVAL;
  constexpr int VEC = 16 / sizeof(T);
  if ((ntaps & 1) == 0 || ntaps < 3 || ntaps > 17 || nx % VEC != 0 || !g_blur3_dma ||
      g_blur3_lxb != kDmaLxb || (reinterpret_cast<uintptr_t>(x) & 15u) ||
      (reinterpret_cast<uintptr_t>(io) & 15u))
    return -2;
  Taps<T> tz, ty, tx;
  bool symmetric = true;
  for (int t = 0; t < kMaxTaps; ++t) {
    tz.w[t] = t < ntaps ? (T)tz_host[t] : T(0);
    ty.w[t] = t < ntaps ? (T)ty_host[t] : T(0);
    tx.w[t] = t < ntaps ? (T)tx_host[t] : T(0);
  }
  for (int t = 0; t < ntaps / 2; ++t)
    symmetric = symmetric && tz.w[t] == tz.w[ntaps - 1 - t] &&
                ty.w[t] == ty.w[ntaps - 1 - t] && tx.w[t] == tx.w[ntaps - 1 - t];
  if (!symmetric) return -2;
  hipStream_t st = as_stream(stream);
#define NSOL_B3E_CASE(N)                                                               \
  case N: return launch_blur3_dma<T, VEC, N, 16, true>(x, io, nz, ny, nx, tz, ty, tx, st, \
                                                        ca, cb, result, ws, ws_doubles);
  switch (ntaps) {
    NSOL_B3E_CASE(3) NSOL_B3E_CASE(5) NSOL_B3E_CASE(7) NSOL_B3E_CASE(9)
    NSOL_B3E_CASE(11) NSOL_B3E_CASE(13) NSOL_B3E_CASE(15) NSOL_B3E_CASE(17)
    default: return -2;
  }
#undef NSOL_B3E_CASE
}

template <typename T, int VEC, int NT>
int launch_blur3(const T *x, T *out, int64_t nz, int64_t ny, int64_t nx,
                 const Taps<T> &tz, const Taps<T> &ty, const Taps<T> &tx, bool symmetric,
                 hipStream_t st) {
  constexpr int NW = 16;
  constexpr int R = NT / 2;
  const int64_t nxv = (nx + VEC - 1) / VEC;
  // lanes per row: a power of two up to 32 that covers the row in few tiles
  int lxb = g_blur3_lxb;
  while (lxb > 8 && lxb / 2 >= nxv) lxb /= 2;
  const int tyr = (NW * 64) / lxb;
  if (tyr < 2 * R) return -2;
  if ((uint64_t)ny * nx * sizeof(T) > kBlur3MaxBytes) return -2;   // 32-bit offsets in a plane
  const int64_t ntx = (nxv + lxb - 1) / lxb;
  const int64_t nty = (ny + tyr - 1) / tyr;
  // z chunks: one 16-wave workgroup runs per CU, so a launch takes
  // ceil(workgroups / CUs) rounds of (chunk + 2R) plane steps; take the chunk
  // count that minimises that (512^3, 64 tiles: 4 chunks = 256 workgroups = one
  // round of 140 steps, 0.418 ms; 8 chunks = two rounds of 76, 0.444 ms)
  const int64_t cus = blur3_cu_count();
  int64_t zchunk = nz, best = -1;
  for (int64_t c = 1; c <= nz && (nz + c - 1) / c >= R; ++c) {
    const int64_t len = (nz + c - 1) / c;
    const int64_t chunks = (nz + len - 1) / len;
    const int64_t rounds = (ntx * nty * chunks + cus - 1) / cus;
    const int64_t cost = rounds * (len + 2 * R);
    if (best < 0 || cost < best) { best = cost; zchunk = len; }
    if (ntx * nty * chunks >= 64 * cus) break;
  }
  if (g_blur3_zchunk > 0) zchunk = g_blur3_zchunk < nz ? g_blur3_zchunk : nz;
  const int64_t nzc = (nz + zchunk - 1) / zchunk;
  const int64_t blocks = ntx * nty * nzc;
  if (blocks > 0x7fffffff) return -2;
  if (g_blur3_dma && symmetric && nx % VEC == 0 && g_blur3_lxb == kDmaLxb) {
    int rc = -2;
    if (g_blur3_nw == 8) rc = launch_blur3_dma<T, VEC, NT, 8>(x, out, nz, ny, nx, tz, ty, tx, st);
    else rc = launch_blur3_dma<T, VEC, NT, 16>(x, out, nz, ny, nx, tz, ty, tx, st);
    if (rc != -2) return rc;
  }
  // two planes per step where the registers allow it
  constexpr int PP = (NT >= 5 && NT <= (sizeof(T) == 4 ? 13 : 11)) ? 2 : 1;
  // (ragged rows: one plane per step, the element-wise edge path needs the registers)
  const int pp = (PP == 2 && nx % VEC != 0) ? 1 : PP;
  const size_t lds = 2 * pp * (size_t)(tyr + 2 * R) * lxb * VEC * sizeof(T);
  if (lds > 150 * 1024) return -2;
  void (*kern)(const T *, T *, int64_t, int64_t, int64_t, Taps<T>, Taps<T>, Taps<T>, int,
               int, int, int);
  if constexpr (PP == 2) {
    if (nx % VEC != 0) kern = k_blur3_wrap_pp<T, VEC, NT, NW, 1, true>;
    else kern = k_blur3_wrap_pp<T, VEC, NT, NW, 2, false>;
  } else {
    if (nx % VEC != 0) return -2;
    kern = k_blur3_wrap<T, VEC, NT, NW>;
  }
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) { (void)hipGetLastError(); return -2; }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NW * 64), lds, st, x, out, nz,
                     ny, nx, tz, ty, tx, lxb, (int)ntx, (int)nty, (int)zchunk);
  return launch_status();
}

// returns -2 when the fused kernel does not apply
template <typename T>
int corr3_impl(const T *x, T *out, int64_t nz, int64_t ny, int64_t nx,
               const double *tz_host, const double *ty_host, const double *tx_host,
               int ntaps, void *stream) {
  if (!x || !out || x == out || !tz_host || !ty_host || !tx_host || nz < 1 ||
      ny < 1 || nx < 1 || ntaps < 1)
    return NSOL_EINVAL;
  constexpr int VEC = 16 / sizeof(T);
  // longer windows do not pay (the z window alone is ntaps vectors per lane)
  constexpr int kMaxFused = 17;   // 0.60 ms against 0.65 for three passes at 512^3; spills beyond
  // (ragged rows: the element-wise wrap of the edge lanes corrects by one period)
  if ((ntaps & 1) == 0 || ntaps < 3 || ntaps > kMaxFused ||
      (nx % VEC != 0 && nx < 2 * ntaps + 2 * VEC) ||
      (reinterpret_cast<uintptr_t>(x) & 15u) || (reinterpret_cast<uintptr_t>(out) & 15u))
    return -2;
  Taps<T> tz, ty, tx;
  for (int t = 0; t < kMaxTaps; ++t) {
    tz.w[t] = t < ntaps ? (T)tz_host[t] : T(0);
    ty.w[t] = t < ntaps ? (T)ty_host[t] : T(0);
    tx.w[t] = t < ntaps ? (T)tx_host[t] : T(0);
  }
  bool symmetric = true;
  for (int t = 0; t < ntaps / 2; ++t)
    symmetric = symmetric && tz.w[t] == tz.w[ntaps - 1 - t] &&
                ty.w[t] == ty.w[ntaps - 1 - t] && tx.w[t] == tx.w[ntaps - 1 - t];
  hipStream_t st = as_stream(stream);
#define NSOL_B3_CASE(N) \
  case N: return launch_blur3<T, VEC, N>(x, out, nz, ny, nx, tz, ty, tx, symmetric, st);
  switch (ntaps) {
    NSOL_B3_CASE(3) NSOL_B3_CASE(5) NSOL_B3_CASE(7) NSOL_B3_CASE(9)
    NSOL_B3_CASE(11) NSOL_B3_CASE(13) NSOL_B3_CASE(15) NSOL_B3_CASE(17)
    default: return -2;
  }
#undef NSOL_B3_CASE
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
  const int rc = try_launch_wrap<T>(x, out, axis, nz, ny, nx, taps, ntaps, centre,
                                    mode, as_stream(stream));
  if (rc != -2) return rc;
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
int nsol_hip_set_param_conv(const char *name, int value) {
  if (!name) return NSOL_EINVAL;
  if (!strcmp(name, "corr_ra")) g_corr_ra = value;
  else if (!strcmp(name, "corr_xv")) g_corr_xv = value;
  else if (!strcmp(name, "corr_blur3_lxb")) g_blur3_lxb = value;
  else if (!strcmp(name, "corr_blur3_zchunk")) g_blur3_zchunk = value;
  else if (!strcmp(name, "corr_blur3_dma")) g_blur3_dma = value;
  else if (!strcmp(name, "corr_blur3_nw")) g_blur3_nw = value;
  else return NSOL_EINVAL;
  return 0;
}
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
int nsol_corr3_wrap_f32(const float *x, float *out, int64_t nz, int64_t ny,
                        int64_t nx, const double *taps_z, const double *taps_y,
                        const double *taps_x, int ntaps, void *stream) {
  return corr3_impl<float>(x, out, nz, ny, nx, taps_z, taps_y, taps_x, ntaps, stream);
}
int nsol_corr3_wrap_f64(const double *x, double *out, int64_t nz, int64_t ny,
                        int64_t nx, const double *taps_z, const double *taps_y,
                        const double *taps_x, int ntaps, void *stream) {
  return corr3_impl<double>(x, out, nz, ny, nx, taps_z, taps_y, taps_x, ntaps, stream);
}
int nsol_corr3_wrap_axpby_f32(const float *x, float *io, int64_t nz, int64_t ny,
                              int64_t nx, const double *taps_z, const double *taps_y,
                              const double *taps_x, int ntaps, double ca, double cb,
                              double *result, double *ws, int64_t ws_doubles,
                              void *stream) {
  return corr3_axpby_impl<float>(x, io, nz, ny, nx, taps_z, taps_y, taps_x, ntaps, ca, cb,
                                 result, ws, ws_doubles, stream);
}
int nsol_corr3_wrap_axpby_f64(const double *x, double *io, int64_t nz, int64_t ny,
                              int64_t nx, const double *taps_z, const double *taps_y,
                              const double *taps_x, int ntaps, double ca, double cb,
                              double *result, double *ws, int64_t ws_doubles,
                              void *stream) {
  return corr3_axpby_impl<double>(x, io, nz, ny, nx, taps_z, taps_y, taps_x, ntaps, ca, cb,
                                  result, ws, ws_doubles, stream);
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
