// Correlation kernels behind the Gaussian blur A = A^T (separable 1-D passes)
// and arbitrary user kernels (dense taps), with scipy.ndimage's boundary modes.
// reference call sites: linear_operators.py:60-68, 82-86 (ndimage.convolve).
#include <string.h>

#include <type_traits>

#include "nsol_common.hpp"
#include "nsol_blur3_dma.hpp"

using namespace nsol;
using namespace nsol_blur3;

namespace {


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

// (nsol_blur3_zchunk -- planes per z chunk of the one-pass blur, 0 = by the round model --
// and nsol_blur3_dma_rag are shared with nsol_blur3_f*.hip: defined below the namespace)
int g_blur3_lxb = 16;   // lanes per row of the one-pass blur's tile (experiment knob)
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
// (rsrc_t, u32x4, u32x2, kNoLane: nsol_blur3_dma.hpp)
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

// nsol_corr3_wrap_axpby_*: io = ca * A x + cb * io in place with the sum of squares
// of the result; -2 when the LDS-DMA kernel does not apply (the caller then blurs
// and combines in two steps)
// nsol_corr3_wrap_norms_* (epi == 2): io = A x stored as it is, result[0] = its sum
// of squares, result[1] = the sum of squares of the weighted forward differences of x
// (ca, cb, cc = the squared weights of d_x, d_y, d_z)
template <typename T>
int corr3_axpby_impl(const T *x, T *io, int64_t nz, int64_t ny, int64_t nx,
                     const double *tz_host, const double *ty_host, const double *tx_host,
                     int ntaps, double ca, double cb, double *result, double *ws,
                     int64_t ws_doubles, void *stream, int epi = 1, double cc = 0.0) {
  if (!x || !io || x == io || !tz_host || !ty_host || !tx_host || !result || !ws ||
      nz < 1 || ny < 1 || nx < 1 || ntaps < 1)
    return NSOL_EINVAL;
  constexpr int VEC = 16 / sizeof(T);
  if ((ntaps & 1) == 0 || ntaps < 3 || ntaps > 17 || !g_blur3_dma ||
      g_blur3_lxb != kDmaLxb || ((reinterpret_cast<uintptr_t>(x) |
                                  reinterpret_cast<uintptr_t>(io)) & (sizeof(T) - 1)))
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
  return blur3_dma_run(x, io, nz, ny, nx, tz, ty, tx, ntaps, epi, ca, cb, cc, result, ws,
                       ws_doubles, st);
}

// nsol_corr3_wrap_lanczos_{init,a,b}_*: the two halves of a Lanczos step on
// A'A + rho B'B taken by the one-pass blur (nsol_blur3_dma.hpp, EPI 3 / 4); -2 when
// that form does not apply (the caller then runs blur, blur and nsol_tk1_lanczos_*)
template <typename T>
bool lanczos_taps(const double *tz_host, const double *ty_host, const double *tx_host,
                  int ntaps, Taps<T> *tz, Taps<T> *ty, Taps<T> *tx, int max_double = 9) {
  // (13 taps in float, 9 in double -- 11 for the lean pair: beyond, a half would spill --
  // nsol_blur3_dma.hpp)
  if ((ntaps & 1) == 0 || ntaps < 5 || ntaps > (sizeof(T) == 4 ? 13 : max_double) || !g_blur3_dma ||
      g_blur3_lxb != kDmaLxb)
    return false;
  for (int t = 0; t < kMaxTaps; ++t) {
    tz->w[t] = t < ntaps ? (T)tz_host[t] : T(0);
    ty->w[t] = t < ntaps ? (T)ty_host[t] : T(0);
    tx->w[t] = t < ntaps ? (T)tx_host[t] : T(0);
  }
  for (int t = 0; t < ntaps / 2; ++t)
    if (tz->w[t] != tz->w[ntaps - 1 - t] || ty->w[t] != ty->w[ntaps - 1 - t] ||
        tx->w[t] != tx->w[ntaps - 1 - t])
      return false;
  return true;
}

template <typename T>
int lanczos_a_impl(const T *y, const T *y_prev, T *t, T *q0, int64_t nz, int64_t ny,
                   int64_t nx, const double *tz_host, const double *ty_host,
                   const double *tx_host, int ntaps, double rho_grad, double rho_ident,
                   double *board, int step, T *coef, double *ws, int64_t ws_doubles,
                   void *stream) {
  if (!y || !t || !q0 || y == t || y == q0 || t == q0 || !tz_host || !ty_host || !tx_host ||
      !board || !coef || !ws || step < 0 || nz < 1 || ny < 1 || nx < 1)
    return NSOL_EINVAL;
  Taps<T> tz, ty, tx;
  if (!lanczos_taps<T>(tz_host, ty_host, tx_host, ntaps, &tz, &ty, &tx)) return -2;
  return blur3_lanczos_a(y, y_prev, t, q0, nz, ny, nx, tz, ty, tx, ntaps, rho_grad, rho_ident,
                         board, step, coef, ws, ws_doubles, as_stream(stream));
}

template <typename T>
int lanczos_b_impl(const T *t, const T *q0, const T *y, T *y_new, int64_t nz, int64_t ny,
                   int64_t nx, const double *tz_host, const double *ty_host,
                   const double *tx_host, int ntaps, double rho_grad, double rho_ident,
                   double *board, int step, T *coef, double *ws, int64_t ws_doubles,
                   void *stream) {
  if (!t || !q0 || !y || !y_new || y_new == t || y_new == q0 || y_new == y || !tz_host ||
      !ty_host || !tx_host || !board || !coef || !ws || step < 0 || nz < 1 || ny < 1 ||
      nx < 1)
    return NSOL_EINVAL;
  Taps<T> tz, ty, tx;
  if (!lanczos_taps<T>(tz_host, ty_host, tx_host, ntaps, &tz, &ty, &tx)) return -2;
  return blur3_lanczos_b(t, q0, y, y_new, nz, ny, nx, tz, ty, tx, ntaps, rho_grad, rho_ident,
                         board, step, coef, ws, ws_doubles, as_stream(stream));
}

// nsol_corr3_wrap_loss_*: A x with the robust-loss data term as its epilogue
// (nsol_blur3_dma.hpp, EPI 5); -2 where that form does not apply (the caller then runs
// the blur and nsol_loss_residual_cost_grad_*)
template <typename T>
int loss_epilogue_impl(const T *x, const T *b, T *g, int64_t nz, int64_t ny, int64_t nx,
                       const double *tz_host, const double *ty_host, const double *tx_host,
                       int ntaps, int loss, double f_scale, double *result, double *ws,
                       int64_t ws_doubles, void *stream) {
  if (!x || !b || !g || x == g || b == g || !tz_host || !ty_host || !tx_host || !result ||
      !ws || nz < 1 || ny < 1 || nx < 1 || loss < 0 || loss > 4 || !(f_scale > 0.0))
    return NSOL_EINVAL;
  if (loss != NSOL_LOSS_LINEAR && loss != NSOL_LOSS_SOFT_L1 && loss != NSOL_LOSS_HUBER)
    return -2;                       // (log1p / atan in double do not fit the blur's registers)
  Taps<T> tz, ty, tx;
  if (!lanczos_taps<T>(tz_host, ty_host, tx_host, ntaps, &tz, &ty, &tx)) return -2;
  return blur3_loss_epilogue(x, b, g, nz, ny, nx, tz, ty, tx, ntaps, loss, f_scale * f_scale,
                             1.345, result, ws, ws_doubles, as_stream(stream));
}

// nsol_corr3_wrap_lanczos_a2_* / _b2_*: the leaner pair of halves -- A is the blur with
// its two sums (EPI 2) landing on the board and closed by a one-thread kernel, B forms the
// step's K'K y itself (EPI 6): no q0 array, 25 B per voxel and step instead of 33
template <typename T>
int lanczos_a2_impl(const T *y, T *t, int64_t nz, int64_t ny, int64_t nx,
                    const double *tz_host, const double *ty_host, const double *tx_host,
                    int ntaps, double rho_grad, double rho_ident, double *board, int step,
                    T *coef, double *ws, int64_t ws_doubles, void *stream) {
  if (!y || !t || y == t || !tz_host || !ty_host || !tx_host || !board || !coef || !ws ||
      step < 0 || nz < 1 || ny < 1 || nx < 1)
    return NSOL_EINVAL;
  Taps<T> tz, ty, tx;
  if (!lanczos_taps<T>(tz_host, ty_host, tx_host, ntaps, &tz, &ty, &tx, 11)) return -2;
  constexpr int VEC = 16 / sizeof(T);
  if (nx % VEC != 0 || ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(t)) & 15u))
    return -2;                                     // (as the other halves: no ragged form)
  // (unit spacing: the squared weights of the difference sums are 1; the sums land on
  // board[3 step + 1], [3 step + 2])
  // (one launch behind the blur: its partials summed onto the board and the second half's
  // coefficients formed -- nsol_blur3::k_blur3_lanczos_final, as after the half with q0)
  return blur3_lanczos_a2(y, t, nz, ny, nx, tz, ty, tx, ntaps, rho_grad, rho_ident, board,
                          step, coef, ws, ws_doubles, as_stream(stream));
}

template <typename T>
int lanczos_b2_impl(const T *t, const T *y, const T *y_prev, T *y_new, int64_t nz, int64_t ny,
                    int64_t nx, const double *tz_host, const double *ty_host,
                    const double *tx_host, int ntaps, double rho_grad, double rho_ident,
                    double *board, int step, T *coef, double *ws, int64_t ws_doubles,
                    void *stream) {
  if (!t || !y || (y_new && (y_new == t || y_new == y || y_new == y_prev)) || !tz_host ||
      !ty_host || !tx_host || !board || !coef || !ws || step < 0 || nz < 1 || ny < 1 || nx < 1)
    return NSOL_EINVAL;
  Taps<T> tz, ty, tx;
  if (!lanczos_taps<T>(tz_host, ty_host, tx_host, ntaps, &tz, &ty, &tx, 11)) return -2;
  return blur3_lanczos_b2(t, y, y_prev, y_new, nz, ny, nx, tz, ty, tx, ntaps, rho_grad,
                          rho_ident, board, step, coef, ws, ws_doubles, as_stream(stream));
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
  if (nsol_blur3_zchunk > 0) zchunk = nsol_blur3_zchunk < nz ? nsol_blur3_zchunk : nz;
  const int64_t nzc = (nz + zchunk - 1) / zchunk;
  const int64_t blocks = ntx * nty * nzc;
  if (blocks > 0x7fffffff) return -2;
  if (g_blur3_dma && symmetric && g_blur3_lxb == kDmaLxb) {
    int rc = -2;
    rc = blur3_dma_run(x, out, nz, ny, nx, tz, ty, tx, NT, 0, 1.0, 0.0, 0.0, nullptr, nullptr,
                       0, st);
    if (rc != -2) return rc;
  }
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15u) return -2;
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
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & (sizeof(T) - 1)))
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
int nsol_blur3_zchunk = 0;
int nsol_blur3_dma_rag = 1;   // LDS-DMA staged blur also for rows / operands off the 16-byte grid
}

extern "C" {
int nsol_hip_set_param_conv(const char *name, int value) {
  if (!name) return NSOL_EINVAL;
  if (!strcmp(name, "corr_ra")) g_corr_ra = value;
  else if (!strcmp(name, "corr_xv")) g_corr_xv = value;
  else if (!strcmp(name, "corr_blur3_lxb")) g_blur3_lxb = value;
  else if (!strcmp(name, "corr_blur3_zchunk")) nsol_blur3_zchunk = value;
  else if (!strcmp(name, "corr_blur3_dma")) g_blur3_dma = value;
  else if (!strcmp(name, "corr_blur3_dma_rag")) nsol_blur3_dma_rag = value;
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
int nsol_corr3_wrap_norms_f32(const float *x, float *out, int64_t nz, int64_t ny,
                              int64_t nx, const double *taps_z, const double *taps_y,
                              const double *taps_x, int ntaps, double wx, double wy,
                              double wz, double *result, double *ws, int64_t ws_doubles,
                              void *stream) {
  return corr3_axpby_impl<float>(x, out, nz, ny, nx, taps_z, taps_y, taps_x, ntaps, wx * wx,
                                 wy * wy, result, ws, ws_doubles, stream, 2, wz * wz);
}
int nsol_corr3_wrap_norms_f64(const double *x, double *out, int64_t nz, int64_t ny,
                              int64_t nx, const double *taps_z, const double *taps_y,
                              const double *taps_x, int ntaps, double wx, double wy,
                              double wz, double *result, double *ws, int64_t ws_doubles,
                              void *stream) {
  return corr3_axpby_impl<double>(x, out, nz, ny, nx, taps_z, taps_y, taps_x, ntaps, wx * wx,
                                  wy * wy, result, ws, ws_doubles, stream, 2, wz * wz);
}
#define NSOL_LANCZOS_DEF(T, SUF)                                                          \
  int nsol_corr3_wrap_lanczos_init_##SUF(double *board, T *coef, double rho_grad,         \
                                         double rho_ident, void *stream) {                \
    if (!board || !coef) return NSOL_EINVAL;                                              \
    return blur3_lanczos_init(board, coef, rho_grad, rho_ident, as_stream(stream));       \
  }                                                                                       \
  int nsol_corr3_wrap_lanczos_a_##SUF(                                                    \
      const T *y, const T *y_prev, T *t, T *q0, int64_t nz, int64_t ny, int64_t nx,       \
      const double *taps_z, const double *taps_y, const double *taps_x, int ntaps,        \
      double rho_grad, double rho_ident, double *board, int step, T *coef, double *ws,    \
      int64_t ws_doubles, void *stream) {                                                 \
    return lanczos_a_impl<T>(y, y_prev, t, q0, nz, ny, nx, taps_z, taps_y, taps_x, ntaps, \
                             rho_grad, rho_ident, board, step, coef, ws, ws_doubles,      \
                             stream);                                                     \
  }                                                                                       \
  int nsol_corr3_wrap_lanczos_b_##SUF(                                                    \
      const T *t, const T *q0, const T *y, T *y_new, int64_t nz, int64_t ny, int64_t nx,  \
      const double *taps_z, const double *taps_y, const double *taps_x, int ntaps,        \
      double rho_grad, double rho_ident, double *board, int step, T *coef, double *ws,    \
      int64_t ws_doubles, void *stream) {                                                 \
    return lanczos_b_impl<T>(t, q0, y, y_new, nz, ny, nx, taps_z, taps_y, taps_x, ntaps,  \
                             rho_grad, rho_ident, board, step, coef, ws, ws_doubles,      \
                             stream);                                                     \
  }                                                                                       \
  int nsol_corr3_wrap_lanczos_a2_##SUF(                                                   \
      const T *y, T *t, int64_t nz, int64_t ny, int64_t nx, const double *taps_z,         \
      const double *taps_y, const double *taps_x, int ntaps, double rho_grad,             \
      double rho_ident, double *board, int step, T *coef, double *ws, int64_t ws_doubles, \
      void *stream) {                                                                     \
    return lanczos_a2_impl<T>(y, t, nz, ny, nx, taps_z, taps_y, taps_x, ntaps, rho_grad,  \
                              rho_ident, board, step, coef, ws, ws_doubles, stream);      \
  }                                                                                       \
  int nsol_corr3_wrap_lanczos_b2_##SUF(                                                   \
      const T *t, const T *y, const T *y_prev, T *y_new, int64_t nz, int64_t ny,          \
      int64_t nx, const double *taps_z, const double *taps_y, const double *taps_x,       \
      int ntaps, double rho_grad, double rho_ident, double *board, int step, T *coef,     \
      double *ws, int64_t ws_doubles, void *stream) {                                     \
    return lanczos_b2_impl<T>(t, y, y_prev, y_new, nz, ny, nx, taps_z, taps_y, taps_x,    \
                              ntaps, rho_grad, rho_ident, board, step, coef, ws,          \
                              ws_doubles, stream);                                        \
  }
NSOL_LANCZOS_DEF(float, f32)
NSOL_LANCZOS_DEF(double, f64)
#define NSOL_LOSS_EPI_DEF(T, SUF)                                                         \
  int nsol_corr3_wrap_loss_##SUF(const T *x, const T *b, T *g, int64_t nz, int64_t ny,    \
                                 int64_t nx, const double *taps_z, const double *taps_y,  \
                                 const double *taps_x, int ntaps, int loss,               \
                                 double f_scale, double *result, double *ws,              \
                                 int64_t ws_doubles, void *stream) {                      \
    return loss_epilogue_impl<T>(x, b, g, nz, ny, nx, taps_z, taps_y, taps_x, ntaps,      \
                                 loss, f_scale, result, ws, ws_doubles, stream);          \
  }
NSOL_LOSS_EPI_DEF(float, f32)
NSOL_LOSS_EPI_DEF(double, f64)
#undef NSOL_LOSS_EPI_DEF
#undef NSOL_LANCZOS_DEF
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
