// Length-n building blocks of the GPU-resident L-BFGS-B driver
// (nsol_amd/lbfgsb.py): projected-gradient norm, Cauchy-point set-up and
// finish, breakpoint selection / gathers, masked dots, limited-memory
// combinations, projected / truncated subspace steps.  They replace the O(n)
// host loops of scipy.optimize's L-BFGS-B behind
// tikhonov_linear_solver.py:197-220.  All are single HBM passes; reductions
// are deterministic two-stage fp64 sums / mins (no float atomics); only the
// breakpoint compaction uses an integer atomic counter (its output is sorted
// on the host, so the result does not depend on arrival order).
#include <math.h>

#include <string.h>

#include "nsol_common.hpp"

using namespace nsol;

namespace {

constexpr int kRed = 1024;   // partial blocks per reduction (ws >= 4*kRed doubles)

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  return v;
}
__device__ __forceinline__ double wmaxd(double v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1)
    v = fmax(v, __shfl_down(v, off, kWave));
  return v;
}

// block reduction of NV values (sum, or max when is_max) -> ws[k*kRed + block]
template <int NV>
__device__ __forceinline__ void block_partials(double (&a)[NV], double *ws,
                                               bool is_max) {
  __shared__ double s[NV][kBlock / kWave];
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const double v = is_max ? wmaxd(a[k]) : wsum(a[k]);
    if (lane == 0) s[k][wv] = v;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    const int k = threadIdx.x;
    double t = s[k][0];
    for (int j = 1; j < kBlock / kWave; ++j)
      t = is_max ? fmax(t, s[k][j]) : t + s[k][j];
    ws[(int64_t)k * kRed + blockIdx.x] = t;
  }
}

// one workgroup: every wave strides over the partials of one statistic after
// the other (fixed order -> deterministic)
__global__ __launch_bounds__(kBlock) void k_final(const double *ws, int nparts,
                                                   int nv, bool is_max,
                                                   double *result) {
  __shared__ double s[kBlock / kWave];
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  for (int k = 0; k < nv; ++k) {
    double t = is_max ? -INFINITY : 0.0;
    for (int j = threadIdx.x; j < nparts; j += kBlock) {
      const double v = ws[(int64_t)k * kRed + j];
      t = is_max ? fmax(t, v) : t + v;
    }
    t = is_max ? wmaxd(t) : wsum(t);
    if (lane == 0) s[wv] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
      double r = s[0];
      for (int j = 1; j < kBlock / kWave; ++j)
        r = is_max ? fmax(r, s[j]) : r + s[j];
      result[k] = r;
    }
    __syncthreads();
  }
}

inline int rgrid(int64_t n) {
  int g = grid_for(n);
  return g > kRed ? kRed : g;
}

#define GRID_STRIDE(i, n)                                                     \
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n);   \
       i += (int64_t)gridDim.x * blockDim.x)

// ---- max |projected gradient| (VEC elements = 16 bytes per lane and trip)
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_projgr(const T *__restrict__ x,
                                                    const T *__restrict__ g,
                                                    int64_t n, T lo, T hi,
                                                    bool has_lo, bool has_hi,
                                                    double *ws) {
  typedef T V __attribute__((ext_vector_type(VEC)));
  double a[1] = {0.0};
  const int64_t nv = n / VEC;
  GRID_STRIDE(j, nv) {
    const V gv = reinterpret_cast<const V *>(g)[j];
    const V xv = reinterpret_cast<const V *>(x)[j];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      T gi = gv[e];
      if (gi < T(0)) {
        if (has_hi) gi = t_max(xv[e] - hi, gi);
      } else {
        if (has_lo) gi = (xv[e] - lo < gi) ? xv[e] - lo : gi;
      }
      a[0] = fmax(a[0], fabs((double)gi));
    }
  }
  block_partials<1>(a, ws, true);
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void k_count_free(const int8_t *iw,
                                                        int64_t n, double *ws) {
  typedef int8_t M __attribute__((ext_vector_type(VEC)));
  double a[1] = {0.0};
  const int64_t nv = n / VEC;
  GRID_STRIDE(j, nv) {
    const M m = reinterpret_cast<const M *>(iw)[j];
    int c = 0;
#pragma unroll
    for (int e = 0; e < VEC; ++e) c += (m[e] <= 0) ? 1 : 0;
    a[0] += (double)c;
  }
  block_partials<1>(a, ws, false);
}

// 16-byte loads when n and the pointers allow (vec = 16 / sizeof(T) elements)
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_mdot(const T *__restrict__ x,
                                                  const T *__restrict__ y,
                                                  const int8_t *iw, int64_t n,
                                                  double *ws) {
  typedef T V __attribute__((ext_vector_type(VEC)));
  typedef int8_t M __attribute__((ext_vector_type(VEC)));
  double a[1] = {0.0};
  const int64_t nv = n / VEC;
  GRID_STRIDE(j, nv) {
    const V xv = reinterpret_cast<const V *>(x)[j];
    const V yv = reinterpret_cast<const V *>(y)[j];
    if (iw) {
      const M m = reinterpret_cast<const M *>(iw)[j];
#pragma unroll
      for (int k = 0; k < VEC; ++k)
        if (m[k] <= 0) a[0] += (double)xv[k] * (double)yv[k];
    } else {
#pragma unroll
      for (int k = 0; k < VEC; ++k) a[0] += (double)xv[k] * (double)yv[k];
    }
  }
  block_partials<1>(a, ws, false);
}

// ---- one vector against several: result[k] = sum_free vecs[k] * v ------------
// (v and the mask are read once instead of once per product)
constexpr int kDotsMax = 24;     // per launch (a 12-vector instantiation serves the short lists)

template <typename T>
struct DotsPtrs {
  const T *p[kDotsMax];
};

template <typename T, int VEC, int NV>
__global__ __launch_bounds__(kBlock) void k_mdots(DotsPtrs<T> P, int nvec,
                                                   const T *__restrict__ y,
                                                   const int8_t *iw, int64_t n,
                                                   double *ws) {
  typedef T V __attribute__((ext_vector_type(VEC)));
  typedef int8_t M __attribute__((ext_vector_type(VEC)));
  double a[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) a[k] = 0.0;
  const int64_t nv = n / VEC;
  GRID_STRIDE(j, nv) {
    V yv = reinterpret_cast<const V *>(y)[j];
    if (iw) {
      const M m = reinterpret_cast<const M *>(iw)[j];
#pragma unroll
      for (int e = 0; e < VEC; ++e) yv[e] = m[e] <= 0 ? yv[e] : T(0);
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      if (k < nvec) {
        const V xv = reinterpret_cast<const V *>(P.p[k])[j];
#pragma unroll
        for (int e = 0; e < VEC; ++e) a[k] += (double)xv[e] * (double)yv[e];
      }
    }
  }
  // per-statistic block sums; layout ws[k * kRed + block] like block_partials
  __shared__ double s[NV][kBlock / kWave];
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const double v = wsum(a[k]);
    if (lane == 0) s[k][wv] = v;
  }
  __syncthreads();
  if (threadIdx.x < nvec) {
    const int k = threadIdx.x;
    double t = s[k][0];
    for (int j = 1; j < kBlock / kWave; ++j) t += s[k][j];
    ws[(int64_t)k * kRed + blockIdx.x] = t;
  }
}

template <typename T>
int mdots_impl(const T *const *vecs, int nvec, const T *y, const int8_t *iwhere,
               int64_t n, double *result, double *ws, void *stream) {
  if (!vecs || nvec < 1 || !y || n < 1 || !result || !ws) return NSOL_EINVAL;
  constexpr int VW = 16 / sizeof(T);
  bool vec = n % VW == 0 && !((uintptr_t)y & 15) &&
             (!iwhere || !((uintptr_t)iwhere & (VW - 1)));
  for (int k = 0; k < nvec; ++k) {
    if (!vecs[k]) return NSOL_EINVAL;
    vec = vec && !((uintptr_t)vecs[k] & 15);
  }
  const int gr = rgrid(vec ? n / VW : n);
  for (int b = 0; b < nvec; b += kDotsMax) {       // at most kDotsMax per launch
    const int cnt = nvec - b < kDotsMax ? nvec - b : kDotsMax;
    DotsPtrs<T> P;
    for (int k = 0; k < kDotsMax; ++k) P.p[k] = k < cnt ? vecs[b + k] : nullptr;
    // (W'v with ten stored pairs is 20 vectors against one: one pass over v
    // instead of two)
    if (vec && cnt > 12)
      hipLaunchKernelGGL((k_mdots<T, VW, kDotsMax>), dim3(gr), dim3(kBlock), 0,
                         as_stream(stream), P, cnt, y, iwhere, n, ws);
    else if (vec)
      hipLaunchKernelGGL((k_mdots<T, VW, 12>), dim3(gr), dim3(kBlock), 0,
                         as_stream(stream), P, cnt, y, iwhere, n, ws);
    else
      hipLaunchKernelGGL((k_mdots<T, 1, kDotsMax>), dim3(gr), dim3(kBlock), 0,
                         as_stream(stream), P, cnt, y, iwhere, n, ws);
    hipLaunchKernelGGL(k_final, dim3(1), dim3(kBlock), 0, as_stream(stream), ws, gr,
                       cnt, false, result + b);
  }
  return launch_status();
}

// out = a - b ; result = { sum out^2, sum out * c } (c may be null): the line
// search's d = z - x with d'd and g'd, and the BFGS pair's y = g - g_old with y'y
// (scipy's lnsrlb / matupd), one pass instead of a difference and two dots.
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_diff_dots(const T *__restrict__ a,
                                                       const T *__restrict__ b,
                                                       const T *__restrict__ c,
                                                       T *__restrict__ out, int64_t n,
                                                       double *ws) {
  typedef T V __attribute__((ext_vector_type(VEC)));
  double acc[2] = {0.0, 0.0};
  const int64_t nv = n / VEC;
  GRID_STRIDE(j, nv) {
    V ov;
    if constexpr (VEC == 1) {
      ov[0] = T(1) * a[j] + T(-1) * b[j];
      out[j] = ov[0];
    } else {
      const V av = reinterpret_cast<const V *>(a)[j];
      const V bv = reinterpret_cast<const V *>(b)[j];
#pragma unroll
      for (int e = 0; e < VEC; ++e) ov[e] = T(1) * av[e] + T(-1) * bv[e];
      reinterpret_cast<V *>(out)[j] = ov;
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[0] += (double)ov[e] * (double)ov[e];
    if (c) {
      V cv;
      if constexpr (VEC == 1) cv[0] = c[j];
      else cv = reinterpret_cast<const V *>(c)[j];
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[1] += (double)ov[e] * (double)cv[e];
    }
  }
  block_partials<2>(acc, ws, false);
}

template <typename T>
int diff_dots_impl(const T *a, const T *b, const T *c, T *out, int64_t n,
                   double *result, double *ws, void *stream) {
  if (!a || !b || !out || n < 1 || !result || !ws) return NSOL_EINVAL;
  constexpr int VW = 16 / sizeof(T);
  const bool vec = n % VW == 0 && !(((uintptr_t)a | (uintptr_t)b | (uintptr_t)out |
                                     (uintptr_t)c) & 15);
  const int gr = rgrid(vec ? n / VW : n);
  if (vec)
    hipLaunchKernelGGL((k_diff_dots<T, VW>), dim3(gr), dim3(kBlock), 0,
                       as_stream(stream), a, b, c, out, n, ws);
  else
    hipLaunchKernelGGL((k_diff_dots<T, 1>), dim3(gr), dim3(kBlock), 0, as_stream(stream),
                       a, b, c, out, n, ws);
  hipLaunchKernelGGL(k_final, dim3(1), dim3(kBlock), 0, as_stream(stream), ws, gr, 2,
                     false, result);
  return launch_status();
}

// ---- masked Gram matrix of up to kGramMax vectors in one pass ---------------
// out[(i,j)], i <= j, = sum over free variables of v_i * v_j, for all pairs at
// once: a workgroup stages a tile of 4 KiB of every vector in LDS
// (zero where the variable is not free), then thread (pair, split) accumulates
// its product over the tile in double.  One read of each vector instead of one
// per pair: the subspace matrix of L-BFGS-B needs 2c^2 + c masked dots of c = 1..m
// stored pairs (tikhonov_linear_solver.py:214-220 -> scipy's formk).
constexpr int kGramMax = 32;
template <typename T> constexpr int gram_tile() { return 4096 / (int)sizeof(T); }   // voxels staged per step
constexpr int kGramBlocks = 512;
constexpr int kGramMaxVec = 24;                 // 8 x 8 blocks of 3 x 3 pairs
constexpr int kGramEnt = 36 * 9;                // block entries one workgroup writes

template <typename T>
struct GramPtrs {
  const T *p[kGramMax];
};

// Register tiling: a thread owns a 3 x 3 block of pairs (vectors 3bi..3bi+2 against
// 3bj..3bj+2, bi <= bj) for a share of the tile's voxels, so six LDS reads and six
// conversions feed nine double FMAs (a thread per single pair needed two reads and
// two conversions per FMA).
constexpr int kGramB = 3;

template <typename T>
__global__ __launch_bounds__(kBlock) void k_masked_gram(GramPtrs<T> P, int nvec,
                                                         const int8_t *iw, int64_t n,
                                                         int nb, int splits, double *ws) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gram_raw[];
  T *tile = reinterpret_cast<T *>(gram_raw);           // [nb * 3][kGramTile + pad]
  constexpr int kGramTile = gram_tile<T>();
  constexpr int VEC = 16 / sizeof(T);
  constexpr int LD = kGramTile + VEC;                   // row pitch: rows on different banks
  typedef T V __attribute__((ext_vector_type(VEC)));
  typedef int8_t M __attribute__((ext_vector_type(VEC)));
  const int tid = threadIdx.x;
  const int nblk = nb * (nb + 1) / 2;
  const int blk = tid / splits, q = tid - blk * splits;
  const bool worker = blk < nblk;
  int bi = 0, bj = 0;
  if (worker) {                                         // blk -> (bi, bj), bi <= bj
    int r = blk;
    while (r >= nb - bi) { r -= nb - bi; ++bi; }
    bj = bi + r;
  }
  double acc[kGramB][kGramB];
#pragma unroll
  for (int u = 0; u < kGramB; ++u)
#pragma unroll
    for (int v = 0; v < kGramB; ++v) acc[u][v] = 0.0;
  // rows beyond nvec (padding of the last block) stay zero
  for (int v = nvec; v < nb * kGramB; ++v)
    for (int e = tid; e < kGramTile; e += kBlock) tile[v * LD + e] = T(0);
  const int64_t ntiles = (n + kGramTile - 1) / kGramTile;
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int64_t base = t * kGramTile;
    // stage: every thread brings VEC consecutive voxels of every vector
    for (int e = tid * VEC; e < kGramTile; e += kBlock * VEC) {
      const int64_t g = base + e;
      const bool full = g + VEC <= n;       // pointers are 16-byte aligned (host check)
      bool keep[VEC];
      if (full && iw) {
        const M m = *reinterpret_cast<const M *>(iw + g);
#pragma unroll
        for (int k = 0; k < VEC; ++k) keep[k] = m[k] <= 0;
      } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k)
          keep[k] = (g + k < n) && (!iw || iw[g + k] <= 0);
      }
      for (int v = 0; v < nvec; ++v) {
        V val;
        if (full) {
          val = *reinterpret_cast<const V *>(P.p[v] + g);
        } else {
#pragma unroll
          for (int k = 0; k < VEC; ++k) val[k] = (g + k < n) ? P.p[v][g + k] : T(0);
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) val[k] = keep[k] ? val[k] : T(0);
        *reinterpret_cast<V *>(tile + v * LD + e) = val;
      }
    }
    __syncthreads();
    if (worker) {
      const T *a = tile + (bi * kGramB) * LD, *b = tile + (bj * kGramB) * LD;
      for (int e = q * VEC; e < kGramTile; e += splits * VEC) {
        V av[kGramB], bv[kGramB];
#pragma unroll
        for (int u = 0; u < kGramB; ++u) {
          av[u] = *reinterpret_cast<const V *>(a + u * LD + e);
          bv[u] = *reinterpret_cast<const V *>(b + u * LD + e);
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          double ad[kGramB], bd[kGramB];
#pragma unroll
          for (int u = 0; u < kGramB; ++u) { ad[u] = (double)av[u][k]; bd[u] = (double)bv[u][k]; }
#pragma unroll
          for (int u = 0; u < kGramB; ++u)
#pragma unroll
            for (int v = 0; v < kGramB; ++v) acc[u][v] += ad[u] * bd[v];
        }
      }
    }
    __syncthreads();
  }
  // sum the splits of every block entry inside the workgroup (fixed order)
  double *red = reinterpret_cast<double *>(gram_raw);   // [kBlock][9]
#pragma unroll
  for (int u = 0; u < kGramB; ++u)
#pragma unroll
    for (int v = 0; v < kGramB; ++v)
      red[tid * (kGramB * kGramB) + u * kGramB + v] = worker ? acc[u][v] : 0.0;
  __syncthreads();
  const int nent = nblk * kGramB * kGramB;
  for (int o = tid; o < nent; o += kBlock) {
    const int kb = o / (kGramB * kGramB), ent = o - kb * (kGramB * kGramB);
    double sum = 0.0;
    for (int sq = 0; sq < splits; ++sq)
      sum += red[(kb * splits + sq) * (kGramB * kGramB) + ent];
    ws[(int64_t)blockIdx.x * kGramEnt + o] = sum;
  }
}

// one workgroup per pair (i <= j): sums the workgroups' partials in a fixed order
__global__ __launch_bounds__(kBlock) void k_gram_final(const double *ws, int nblocks,
                                                        int nvec, int nb,
                                                        double *result) {
  __shared__ double s[kBlock / kWave];
  int vi = 0, r = blockIdx.x;
  while (r >= nvec - vi) { r -= nvec - vi; ++vi; }
  const int vj = vi + r;
  const int bi = vi / kGramB, bj = vj / kGramB;
  int kb = bj - bi;                                 // block index of (bi, bj), bi <= bj
  for (int u = 0; u < bi; ++u) kb += nb - u;
  const int o = kb * (kGramB * kGramB) + (vi % kGramB) * kGramB + (vj % kGramB);
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  double t = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += kBlock) t += ws[(int64_t)b * kGramEnt + o];
  t = wsum(t);
  if (lane == 0) s[wv] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    double rr = s[0];
    for (int j = 1; j < kBlock / kWave; ++j) rr += s[j];
    result[blockIdx.x] = rr;
  }
}

// ---- the same Gram matrix with the tiles staged by LDS-DMA -------------------
// k_masked_gram above runs one 4-wave workgroup per CU at ten stored pairs (its
// 86 KiB tile is LDS-limited) and alternates between "load a tile" and "multiply
// it": 9.2 ms at 512^3 for 1.9 ms worth of traffic.  Here a 16-wave workgroup keeps
// THREE tiles in LDS (TB bytes per vector and tile: 8 KiB up to 6 vectors, 4 KiB
// up to 12, 2 KiB up to 24 -- about 144 KiB in all): the tiles of steps t + 1 and
// t + 2 travel global -> LDS with global_load_lds_dwordx4 (no registers; a counted
// s_waitcnt vmcnt(N) leaves the newer one in flight across the barrier) while step
// t is multiplied; a thread owns a 4 x 4 block of pairs for a share of the tile, so
// eight LDS reads feed sixteen double FMAs per element; the free-variable mask is
// staged the same way and applied to one side of the products.  Same sums in a
// different order as k_masked_gram (both fixed; double accumulation).
constexpr int kGram2B = 4;                       // pair block edge
constexpr int kGram2Threads = 1024;
constexpr int kGram2Waves = kGram2Threads / kWave;
constexpr int kGram2Bufs = 3;
constexpr int kGram2MaxVec = 24;                 // 6 x 6 blocks -> 21 block pairs
constexpr int kGram2Ent = 21 * kGram2B * kGram2B;

// RG: the same pass also forms the reduced gradient of the subspace step,
//   r = free ? b0 * base0 + b1 * base1 + b2 * base2 + sum_k wc[k] * vec_k : 0
// (scipy's cmprlb: -theta (xcp - x) - g + W M c; what k_wcomb computes from a pass
// of its own over the same 2c vectors, term for term in the same order): the three
// base vectors are staged like three more rows of the tile, and the first lanes
// combine the tile's rows once they have landed.
template <typename T>
struct GramRG {
  const T *base[3];
  T bcoef[3];
  T wcoef[kGram2MaxVec];
  T *out;
};

template <typename T, int TB, bool RG = false>
__global__ __launch_bounds__(kGram2Threads) void k_masked_gram_dma(
    GramPtrs<T> P, int nvec, const int8_t *iw, int64_t n, int nb, int splits, double *ws,
    GramRG<T> R = GramRG<T>()) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gram_raw[];
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int kTile = TB / (int)sizeof(T);                   // voxels per tile
  constexpr int kQuads = kTile / VEC;                          // 16-byte groups per row
  constexpr int kPitch = TB + 16;                              // bytes between rows
  constexpr int kMaskBytes = kTile;                            // int8 per voxel
  constexpr int kRowPieces = TB / 1024;                        // 1 KiB per wave-instruction
  constexpr int kMaskPieces = (kMaskBytes + 1023) / 1024;
  typedef T V __attribute__((ext_vector_type(VEC)));
  const int rows = nb * kGram2B;
  constexpr int kExtra = RG ? 3 : 0;                           // base vectors of r
  const int buf_bytes = (rows + kExtra) * kPitch + ((kMaskBytes + 15) & ~15);
  const int mask_at = (rows + kExtra) * kPitch;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nblk = nb * (nb + 1) / 2;
  const int blk = tid / splits, q = tid - blk * splits;
  const bool worker = blk < nblk;
  int bi = 0, bj = 0;
  if (worker) {                                         // blk -> (bi, bj), bi <= bj
    int r = blk;
    while (r >= nb - bi) { r -= nb - bi; ++bi; }
    bj = bi + r;
  }
  double acc[kGram2B][kGram2B];
#pragma unroll
  for (int u = 0; u < kGram2B; ++u)
#pragma unroll
    for (int v = 0; v < kGram2B; ++v) acc[u][v] = 0.0;
  // (RG) The owners of the DIAGONAL pair blocks leave out the lower triangle nobody
  // reads and use four of its slots -- (1,0), (2,0), (2,1), (3,0) for the block's vectors
  // 0 .. 3 -- for the products of their four vectors with
  //   b = b0 * base0 + b1 * base1 + b2 * base2   (r without its W part):
  // W'Z r = W'Z b + (W'Z W) wc then follows on the host from this pass's matrix, and the
  // pass over the 2c vectors that formed W'Z r on its own (nsol_lb_mdots_*) is not needed.
  // (Accumulators of their own cost eight registers this kernel does not have.)
  const bool diag = RG && worker && bi == bj;
  // rows beyond nvec (padding of the last block) stay zero in every buffer
  for (int b = 0; b < kGram2Bufs; ++b)
    for (int v = nvec; v < rows; ++v)
      for (int e = tid * 16; e < TB; e += kGram2Threads * 16)
        *reinterpret_cast<uint4 *>(gram_raw + b * buf_bytes + v * kPitch + e) =
            make_uint4(0, 0, 0, 0);
  const int64_t ntiles = (n + kTile - 1) / kTile;
  // pieces of one tile: vector v, part h (1 KiB each), then the mask; piece k
  // belongs to wave k % kGram2Waves
  const int nsrc = nvec + kExtra;                             // staged vectors
  const int npieces = kRowPieces * nsrc + (iw ? kMaskPieces : 0);
  const int my_pieces = (npieces - wave + kGram2Waves - 1) / kGram2Waves;   // wave-uniform
  auto stage = [&](int64_t t, int b) {
    const int64_t base = t * kTile;
    unsigned char *dst = gram_raw + b * buf_bytes;
    for (int k = wave; k < npieces; k += kGram2Waves) {
      if (k < kRowPieces * nsrc) {
        const int v = k / kRowPieces, h = k - v * kRowPieces;
        const int64_t e = base + (int64_t)h * (1024 / (int)sizeof(T)) + (int64_t)lane * VEC;
        const T *src = v < nvec ? P.p[v] : R.base[v < nvec ? 0 : v - nvec];
        const int lrow = v < nvec ? v : rows + (v - nvec);
        if (e < n)                                       // n % VEC == 0 (host check)
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void *)(src + e),
              (__attribute__((address_space(3))) void *)(dst + lrow * kPitch + h * 1024),
              16, 0, 0);
      } else {
        const int h = k - kRowPieces * nsrc;
        const int64_t e = base + (int64_t)h * 1024 + (int64_t)lane * 16;
        if (h * 1024 + lane * 16 < kMaskBytes && e < n)   // n % 16 == 0 with a mask
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void *)(iw + e),
              (__attribute__((address_space(3))) void *)(dst + mask_at + h * 1024),
              16, 0, 0);
      }
    }
  };
  // a tile whose every piece is issued by every lane-complete instruction: only
  // then is the number of this wave's operations in flight known
  auto full = [&](int64_t t) { return (t + 1) * kTile <= n; };
  // (wave-uniform) barrier that leaves `newer` of this wave's loads in flight
  auto sync_keep = [&](int newer) {
    switch (newer) {
      case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      case 7: asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    }
  };
  int64_t t = blockIdx.x;
  const int64_t step = gridDim.x;
  if (t < ntiles) stage(t, 0);
  if (t + step < ntiles) stage(t + step, 1);
  // tile t must have landed; tile t + step may stay in flight if it is a full one
  // (every piece then is one issued instruction: the count is known)
  sync_keep((t + step < ntiles && full(t + step)) ? my_pieces : 0);
  int cur = 0;
  for (; t < ntiles; t += step) {
    const int64_t t2 = t + 2 * step;
    int nxt2 = cur + 2; if (nxt2 >= kGram2Bufs) nxt2 -= kGram2Bufs;
    if (t2 < ntiles) stage(t2, nxt2);
    int stored = 0;                                    // (wave-uniform) stores of r issued
    if (RG && R.out != nullptr) {
      // One 16-byte group of r per lane and trip (k_wcomb's sum, same order), by the
      // waves that own no pair block -- 15 block pairs x 64 shares leave the sixteenth
      // wave free at ten stored pairs.  (On the first lanes instead, two waves did this
      // on top of their pair blocks and the other fourteen waited at the barrier:
      // 3.8 ms against 2.9 for the matrix alone.)
      const int64_t base = t * kTile;
      const int64_t left = (n - base) / VEC;            // groups of this tile in range
      const int quads_here = left < kQuads ? (int)left : kQuads;
      const int w0 = (nblk * splits + kWave - 1) / kWave;          // first free wave
      const int first = w0 < kGram2Waves ? w0 * kWave : 0;
      const int share = w0 < kGram2Waves ? kGram2Threads - first : kGram2Threads;
      const int mine = (wave * kWave - first);                      // this wave's first group
      if (mine >= 0 && mine < quads_here) stored = (quads_here - mine + share - 1) / share;
      for (int e = tid - first; e >= 0 && e < quads_here; e += share) {
        const unsigned char *bufp = gram_raw + cur * buf_bytes;
        V accv = V(T(0));
#pragma unroll
        for (int k = 0; k < 3; ++k)
          accv += R.bcoef[k] * *reinterpret_cast<const V *>(bufp + (rows + k) * kPitch + e * 16);
        // four rows at a time: four LDS reads in flight (one after the other, the 23
        // dependent reads of a group were most of a tile's time); the rows beyond nvec
        // hold zeros and carry zero coefficients
        for (int k = 0; k < rows; k += 4) {
          const V a0 = *reinterpret_cast<const V *>(bufp + k * kPitch + e * 16);
          const V a1 = *reinterpret_cast<const V *>(bufp + (k + 1) * kPitch + e * 16);
          const V a2 = *reinterpret_cast<const V *>(bufp + (k + 2) * kPitch + e * 16);
          const V a3 = *reinterpret_cast<const V *>(bufp + (k + 3) * kPitch + e * 16);
          accv += R.wcoef[k] * a0;
          accv += R.wcoef[k + 1] * a1;
          accv += R.wcoef[k + 2] * a2;
          accv += R.wcoef[k + 3] * a3;
        }
        if (iw) {
          const unsigned char *mk = bufp + mask_at;
#pragma unroll
          for (int k = 0; k < VEC; ++k)
            if ((int8_t)mk[e * VEC + k] > 0) accv[k] = T(0);
        }
        *reinterpret_cast<V *>(R.out + base + (int64_t)e * VEC) = accv;
      }
    }
    if (worker) {
      const unsigned char *bufp = gram_raw + cur * buf_bytes;
      const unsigned char *a = bufp + (bi * kGram2B) * kPitch;
      const unsigned char *bb = bufp + (bj * kGram2B) * kPitch;
      const unsigned char *mk = bufp + mask_at;
      const int64_t base = t * kTile;
      for (int e = q; e < kQuads; e += splits) {
        if (base + (int64_t)e * VEC >= n) break;
        V av[kGram2B], bv[kGram2B];
#pragma unroll
        for (int u = 0; u < kGram2B; ++u) {
          av[u] = *reinterpret_cast<const V *>(a + u * kPitch + e * 16);
          bv[u] = *reinterpret_cast<const V *>(bb + u * kPitch + e * 16);
        }
        bool keep[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) keep[k] = !iw || (int8_t)mk[e * VEC + k] <= 0;
        V bq = V(T(0));
        if constexpr (RG) {
          if (diag) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
              bq += R.bcoef[k] * *reinterpret_cast<const V *>(bufp + (rows + k) * kPitch + e * 16);
          }
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          double ad[kGram2B], bd[kGram2B];
#pragma unroll
          for (int u = 0; u < kGram2B; ++u) {
            ad[u] = (double)(keep[k] ? av[u][k] : T(0));   // (select before widening)
            bd[u] = (double)bv[u][k];
          }
          // one fused multiply-add per product (the library is built without
          // contraction): for float data the product of two widened values is exact
          // in double, so this is bit for bit the multiply and add of k_masked_gram
          if (RG && diag) {
            const double bb = (double)bq[k];
#pragma unroll
            for (int u = 0; u < kGram2B; ++u)
#pragma unroll
              for (int v = u; v < kGram2B; ++v) acc[u][v] = __builtin_fma(ad[u], bd[v], acc[u][v]);
            acc[1][0] = __builtin_fma(ad[0], bb, acc[1][0]);
            acc[2][0] = __builtin_fma(ad[1], bb, acc[2][0]);
            acc[2][1] = __builtin_fma(ad[2], bb, acc[2][1]);
            acc[3][0] = __builtin_fma(ad[3], bb, acc[3][0]);
          } else {
#pragma unroll
            for (int u = 0; u < kGram2B; ++u)
#pragma unroll
              for (int v = 0; v < kGram2B; ++v) acc[u][v] = __builtin_fma(ad[u], bd[v], acc[u][v]);
          }
        }
      }
    }
    // tile t + step must have landed before the next step reads it; the pieces of
    // tile t + 2 step (issued above, the youngest operations of this wave) may stay
    // in flight when their number is known
    // (a wave that stored a piece of r has one more operation behind the pieces)
    sync_keep((t2 < ntiles && full(t2)) ? my_pieces + stored : 0);
    if (++cur == kGram2Bufs) cur = 0;
  }
  // sum the splits of every block entry inside the workgroup (fixed order)
  double *red = reinterpret_cast<double *>(gram_raw);   // [threads][16]
  constexpr int E = kGram2B * kGram2B;
#pragma unroll
  for (int u = 0; u < kGram2B; ++u)
#pragma unroll
    for (int v = 0; v < kGram2B; ++v)
      red[tid * E + u * kGram2B + v] = worker ? acc[u][v] : 0.0;
  __syncthreads();
  const int nent = nblk * E;
  for (int o = tid; o < nent; o += kGram2Threads) {
    const int kb = o / E, ent = o - kb * E;
    double sum = 0.0;
    for (int sq = 0; sq < splits; ++sq) sum += red[(kb * splits + sq) * E + ent];
    ws[(int64_t)blockIdx.x * kGram2Ent + o] = sum;
  }
}

// one workgroup per pair (i <= j): sums the workgroups' partials in a fixed order
__global__ __launch_bounds__(kBlock) void k_gram2_final(const double *ws, int nblocks,
                                                         int nvec, int nb, double *result) {
  __shared__ double s[kBlock / kWave];
  const int npairs = nvec * (nvec + 1) / 2;
  int o;
  if ((int)blockIdx.x >= npairs) {
    // (RG) product of vector k with b: a lower-triangle slot of k's diagonal block
    const int k = (int)blockIdx.x - npairs, d = k / kGram2B, u = k % kGram2B;
    int kb = 0;
    for (int q2 = 0; q2 < d; ++q2) kb += nb - q2;
    const int slot = u == 0 ? 1 * kGram2B : (u == 1 ? 2 * kGram2B : (u == 2 ? 2 * kGram2B + 1
                                                                           : 3 * kGram2B));
    o = kb * (kGram2B * kGram2B) + slot;
  } else {
  int vi = 0, r = blockIdx.x;
  while (r >= nvec - vi) { r -= nvec - vi; ++vi; }
  const int vj = vi + r;
  const int bi = vi / kGram2B, bj = vj / kGram2B;
  int kb = bj - bi;                                 // block index of (bi, bj), bi <= bj
  for (int u = 0; u < bi; ++u) kb += nb - u;
  o = kb * (kGram2B * kGram2B) + (vi % kGram2B) * kGram2B + (vj % kGram2B);
  }
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  double t = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += kBlock) t += ws[(int64_t)b * kGram2Ent + o];
  t = wsum(t);
  if (lane == 0) s[wv] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    double rr = s[0];
    for (int j = 1; j < kBlock / kWave; ++j) rr += s[j];
    result[blockIdx.x] = rr;
  }
}

// ---- the same pass with the products on the matrix cores ---------------------
// k_masked_gram_dma is bound by its float64 VALU work and its LDS reads together (per
// four voxels and ten stored pairs: 480 conversions, 960 FMAs, 120 ds_read_b128; the
// two do not overlap across the barrier of a 512-voxel tile): 3.3 ms at 512^3 where the
// bytes take 2.  v_mfma_f64_4x4x4_4b_f64 multiplies FOUR independent 4 x 4 x 4 blocks per
// instruction (gfx950: 8 ns per instruction and SIMD, 65 TFLOP/s; v_mfma_f64_16x16x4
// sustains 45 and would pad 20 vectors to 32).  Here the four blocks are four groups of
// four VOXELS of one pair of vector blocks: lane l supplies element (l & 3) of a vector
// block at voxel (l >> 2) of a 16-voxel group -- the same register serves as the A
// operand (row i = l & 3) and as the B operand (column j = l & 3), operand layout
//   block (l >> 2) & 3, index l & 3, k = l >> 4          (tools/_probe/mfma_f64_4x4_probe.hip)
// -- so a wave reads NB values per lane and group (one ds_read_b32 / _b64 each, widened
// and masked ONCE) and issues NB (NB + 1) / 2 MFMAs on them; the result register of lane
// l is entry (i = l >> 4, j = l & 3) of the pair block for its voxel quarter (l >> 2) & 3.
// Every wave takes whole groups of every tile (no pair-block owners, no idle lanes);
// staging, the reduced gradient (RG) and the workspace layout are k_masked_gram_dma's.
// Products of widened floats are exact in double, so the sums differ from the VALU
// form only in their (fixed) order.
// measurement builds (tools/_probe/build_variant_lb.sh -DGRAM_ABLATE=n): 1 no products,
// 2 no reduced gradient, 4 no staging.  At 512^3 and ten stored pairs (gram / gram + r, ms,
// same box): 2.30 / 2.72 as built; staging alone 1.72 / 2.02; arithmetic alone 1.50 / 2.09.
#ifndef GRAM_ABLATE
#define GRAM_ABLATE 0
#endif
template <typename T, int TB, int NB, bool RG>
__global__ __launch_bounds__(kGram2Threads) void k_masked_gram_mfma(
    GramPtrs<T> P, int nvec, const int8_t *iw, int64_t n, double *ws,
    GramRG<T> R = GramRG<T>()) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gram_raw[];
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int kTile = TB / (int)sizeof(T);                   // voxels per tile
  constexpr int kQuads = kTile / VEC;                          // 16-byte groups per row
  // rows (l & 3) of a block on different banks for the element reads: ds_read_b32 banks
  // are (a / 4) % 32 per 32-lane half (8 voxels x 4 rows), ds_read_b64's (a / 4) % 64
  constexpr int kPitch = TB + (sizeof(T) == 4 ? 32 : 64);      // bytes between rows
  constexpr int kMaskBytes = kTile;                            // int8 per voxel
  constexpr int kRowPieces = TB / 1024;                        // 1 KiB per wave-instruction
  constexpr int kMaskPieces = (kMaskBytes + 1023) / 1024;
  constexpr int rows = NB * kGram2B;
  constexpr int kExtra = RG ? 3 : 0;                           // base vectors of r
  constexpr int mask_at = (rows + kExtra) * kPitch;
  constexpr int buf_bytes = mask_at + ((kMaskBytes + 15) & ~15);
  constexpr int NP = NB * (NB + 1) / 2;                        // pair blocks
  constexpr int kGroups = kTile / 16;                          // 16-voxel groups per tile
  constexpr int kRWaves = kTile / kWave;                       // (RG) waves that form r
  static_assert(kGroups % kGram2Waves == 0 && (!RG || (kRWaves >= 1 && kRWaves <= kGram2Waves)), "tile");
  typedef T V __attribute__((ext_vector_type(VEC)));
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r4 = lane & 3, vox = lane >> 2;
  double acc[NP], accb[NB];
#pragma unroll
  for (int p = 0; p < NP; ++p) acc[p] = 0.0;
#pragma unroll
  for (int X = 0; X < NB; ++X) accb[X] = 0.0;
  // rows beyond nvec (padding of the last block) stay zero in every buffer
  for (int b = 0; b < kGram2Bufs; ++b)
    for (int v = nvec; v < rows; ++v)
      for (int e = tid * 16; e < TB; e += kGram2Threads * 16)
        *reinterpret_cast<uint4 *>(gram_raw + b * buf_bytes + v * kPitch + e) =
            make_uint4(0, 0, 0, 0);
  const int64_t ntiles = (n + kTile - 1) / kTile;
  const int nsrc = nvec + kExtra;                             // staged vectors
  const int npieces = kRowPieces * nsrc + (iw ? kMaskPieces : 0);
  const int my_pieces = (npieces - wave + kGram2Waves - 1) / kGram2Waves;   // wave-uniform
  auto stage = [&](int64_t t, int b) {
    const int64_t base = t * kTile;
    unsigned char *dst = gram_raw + b * buf_bytes;
    for (int k = wave; k < npieces; k += kGram2Waves) {
      if (k < kRowPieces * nsrc) {
        const int v = k / kRowPieces, h = k - v * kRowPieces;
        const int64_t e = base + (int64_t)h * (1024 / (int)sizeof(T)) + (int64_t)lane * VEC;
        const T *src = v < nvec ? P.p[v] : R.base[v < nvec ? 0 : v - nvec];
        const int lrow = v < nvec ? v : rows + (v - nvec);
        if (e < n)                                       // n % VEC == 0 (host check)
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void *)(src + e),
              (__attribute__((address_space(3))) void *)(dst + lrow * kPitch + h * 1024),
              16, 0, 0);
      } else {
        const int h = k - kRowPieces * nsrc;
        const int64_t e = base + (int64_t)h * 1024 + (int64_t)lane * 16;
        if (h * 1024 + lane * 16 < kMaskBytes && e < n)   // n % 16 == 0 with a mask
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void *)(iw + e),
              (__attribute__((address_space(3))) void *)(dst + mask_at + h * 1024),
              16, 0, 0);
      }
    }
  };
  auto full = [&](int64_t t) { return (t + 1) * kTile <= n; };
  auto sync_keep = [&](int newer) {
    switch (newer) {
      case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      case 7: asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    }
  };
  int64_t t = blockIdx.x;
  const int64_t step = gridDim.x;
  if (t < ntiles) stage(t, 0);
  if (t + step < ntiles) stage(t + step, 1);
  sync_keep((t + step < ntiles && full(t + step)) ? my_pieces : 0);
  int cur = 0;
  for (; t < ntiles; t += step) {
    const int64_t t2 = t + 2 * step;
    int nxt2 = cur + 2; if (nxt2 >= kGram2Bufs) nxt2 -= kGram2Bufs;
    if (t2 < ntiles && !(GRAM_ABLATE & 4)) stage(t2, nxt2);
    const unsigned char *bufp = gram_raw + cur * buf_bytes;
    const int64_t base = t * kTile;
    int stored = 0;                                    // (wave-uniform) stores of r issued
    if (RG && !(GRAM_ABLATE & 2) && R.out != nullptr) {   // (R.out null: the caller forms r later)
      // one element of r per lane (k_wcomb's sum, same order: the rows beyond nvec hold
      // zeros and carry zero coefficients), by the last kRWaves waves on top of their
      // groups -- element-wise so that half the waves share the work (two waves with a
      // 16-byte group per lane held the other fourteen at the barrier)
      constexpr int first = (kGram2Waves - kRWaves) * kWave;
      const int e = tid - first;                                    // voxel of the tile
      if (wave >= kGram2Waves - kRWaves && base + (wave * kWave - first) < n) stored = 1;
      if (e >= 0 && base + e < n) {
        const unsigned char *col = bufp + e * (int)sizeof(T);
        T accv = T(0);
#pragma unroll
        for (int k = 0; k < 3; ++k)
          accv += R.bcoef[k] * *reinterpret_cast<const T *>(col + (rows + k) * kPitch);
#pragma unroll
        for (int k = 0; k < rows; ++k)
          accv += R.wcoef[k] * *reinterpret_cast<const T *>(col + k * kPitch);
        if (iw && (int8_t)bufp[mask_at + e] > 0) accv = T(0);
        R.out[base + e] = accv;
      }
    }
    if constexpr (!(GRAM_ABLATE & 1)) {
      constexpr int G = kGroups / kGram2Waves;             // groups of this wave per tile
      const unsigned char *mine = bufp + r4 * kPitch + (wave * 16 + vox) * (int)sizeof(T);
      const unsigned char *mk = bufp + mask_at + wave * 16 + vox;
      // every LDS read of the tile first (one wait), then the products
      T vals[G][NB], braw[G][3];
      int8_t mb[G];
#pragma unroll
      for (int gi = 0; gi < G; ++gi) {
        constexpr int kGroupStride = kGram2Waves * 16;     // voxels between this wave's groups
#pragma unroll
        for (int X = 0; X < NB; ++X)
          vals[gi][X] = *reinterpret_cast<const T *>(mine + X * kGram2B * kPitch +
                                                     gi * kGroupStride * (int)sizeof(T));
        if constexpr (RG) {
#pragma unroll
          for (int k = 0; k < 3; ++k)
            braw[gi][k] = *reinterpret_cast<const T *>(
                bufp + (rows + k) * kPitch + (wave * 16 + vox + gi * kGroupStride) * (int)sizeof(T));
        }
        mb[gi] = iw ? (int8_t)mk[gi * kGroupStride] : (int8_t)0;
      }
#pragma unroll
      for (int gi = 0; gi < G; ++gi) {
        const int g0 = (wave + gi * kGram2Waves) * 16;     // first voxel of the group
        // (what lies beyond the end of the vectors is stale LDS)
        const bool keep = (base + g0 + vox < n) & (mb[gi] <= 0);
        double a[NB];
#pragma unroll
        for (int X = 0; X < NB; ++X) a[X] = (double)(keep ? vals[gi][X] : T(0));
        if constexpr (RG) {
          T bq = T(0);
#pragma unroll
          for (int k = 0; k < 3; ++k) bq += R.bcoef[k] * braw[gi][k];
          // (it must not meet the zeros of the masked side as a NaN)
          const double bb = (double)(keep ? bq : T(0));
#pragma unroll
          for (int X = 0; X < NB; ++X) accb[X] = __builtin_fma(a[X], bb, accb[X]);
        }
        int p = 0;
#pragma unroll
        for (int bi = 0; bi < NB; ++bi)
#pragma unroll
          for (int bj = bi; bj < NB; ++bj, ++p)
            acc[p] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[bi], a[bj], acc[p], 0, 0, 0);
      }
    }
    sync_keep((t2 < ntiles && full(t2) && !(GRAM_ABLATE & 4)) ? my_pieces + stored : 0);
    if (++cur == kGram2Bufs) cur = 0;
  }
  // the four voxel quarters of every entry, then the waves, in a fixed order
  double *red = reinterpret_cast<double *>(gram_raw);   // [waves][NP * 16]
  constexpr int E = kGram2B * kGram2B;
  {
    const int i = lane >> 4, j = lane & 3, quarter = (lane >> 2) & 3;
    int p = 0;
#pragma unroll
    for (int bi = 0; bi < NB; ++bi)
#pragma unroll
      for (int bj = bi; bj < NB; ++bj, ++p) {
        double x = acc[p];
        x += __shfl_xor(x, 4);
        x += __shfl_xor(x, 8);
        // (RG) the lower triangle of a diagonal block, which nobody reads, carries the
        // products of the block's vectors with b instead (k_gram2_final's slots)
        if (quarter == 0 && !(RG && bi == bj && i > j)) red[wave * (NP * E) + p * E + i * kGram2B + j] = x;
      }
    if constexpr (RG) {
      int pd = 0;
#pragma unroll
      for (int X = 0; X < NB; ++X) {
        double x = accb[X];
        x += __shfl_xor(x, 4);
        x += __shfl_xor(x, 8);
        x += __shfl_xor(x, 16);
        x += __shfl_xor(x, 32);
        const int slot = lane == 0 ? 1 * kGram2B : (lane == 1 ? 2 * kGram2B
                                                  : (lane == 2 ? 2 * kGram2B + 1 : 3 * kGram2B));
        if (lane < 4) red[wave * (NP * E) + pd * E + slot] = x;
        pd += NB - X;
      }
    }
  }
  __syncthreads();
  for (int o = tid; o < NP * E; o += kGram2Threads) {
    double sum = 0.0;
    for (int w = 0; w < kGram2Waves; ++w) sum += red[w * (NP * E) + o];
    ws[(int64_t)blockIdx.x * kGram2Ent + o] = sum;
  }
}

int g_gram_mfma = 1;             // 1: k_masked_gram_mfma in place of k_masked_gram_dma

template <typename T, int TB, int NB, bool RG>
int masked_gram_mfma_launch(const GramPtrs<T> &P, int nvec, const int8_t *iwhere, int64_t n,
                            double *result, double *ws, hipStream_t st, const GramRG<T> &R) {
  constexpr int kTile = TB / (int)sizeof(T);
  constexpr int kPitch = TB + (sizeof(T) == 4 ? 32 : 64);
  constexpr size_t lds = (size_t)kGram2Bufs *
      ((size_t)(NB * kGram2B + (RG ? 3 : 0)) * kPitch + ((kTile + 15) & ~15));
  static_assert(lds <= 160 * 1024, "three tiles in LDS");
  static_assert(lds >= (size_t)kGram2Waves * (NB * (NB + 1) / 2) * 16 * sizeof(double), "reduction");
  const int64_t ntiles = (n + kTile - 1) / kTile;
  int64_t blocks = 256;                              // one 16-wave workgroup per CU
  if (blocks > kGramBlocks) blocks = kGramBlocks;
  if (blocks > ntiles) blocks = ntiles;
  auto kern = k_masked_gram_mfma<T, TB, NB, RG>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { (void)hipGetLastError(); return -2; }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kGram2Threads), lds, st, P, nvec,
                     iwhere, n, ws, R);
  hipLaunchKernelGGL(k_gram2_final, dim3(nvec * (nvec + 1) / 2 + (RG ? nvec : 0)),
                     dim3(kBlock), 0, st, ws, (int)blocks, nvec, NB, result);
  return launch_status();
}

// tile size by the rows to stage (k_masked_gram_dma's rule)
template <typename T, bool RG>
int masked_gram_mfma_dispatch(const GramPtrs<T> &P, int nvec, const int8_t *iwhere, int64_t n,
                              double *result, double *ws, hipStream_t st, const GramRG<T> &R) {
  const int nb = (nvec + kGram2B - 1) / kGram2B;
  switch (nb) {
    case 1: return masked_gram_mfma_launch<T, RG ? 4096 : 8192, 1, RG>(P, nvec, iwhere, n, result, ws, st, R);
    case 2: return masked_gram_mfma_launch<T, 4096, 2, RG>(P, nvec, iwhere, n, result, ws, st, R);
    case 3: return masked_gram_mfma_launch<T, RG ? 2048 : 4096, 3, RG>(P, nvec, iwhere, n, result, ws, st, R);
    case 4: return masked_gram_mfma_launch<T, 2048, 4, RG>(P, nvec, iwhere, n, result, ws, st, R);
    case 5: return masked_gram_mfma_launch<T, 2048, 5, RG>(P, nvec, iwhere, n, result, ws, st, R);
    case 6: if constexpr (!RG) return masked_gram_mfma_launch<T, 2048, 6, RG>(P, nvec, iwhere, n, result, ws, st, R);
  }
  return -2;
}

int g_gram_dma = 1;              // 1: k_masked_gram_dma where it applies; 0: k_masked_gram

template <typename T, int TB, bool RG>
int masked_gram_dma_launch_tb(const GramPtrs<T> &P, int nvec, const int8_t *iwhere,
                              int64_t n, double *result, double *ws, hipStream_t st,
                              const GramRG<T> &R) {
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int kTile = TB / (int)sizeof(T);
  const int nb = (nvec + kGram2B - 1) / kGram2B;
  const int nblk = nb * (nb + 1) / 2;
  int splits = 1;
  while (splits * 2 * nblk <= kGram2Threads && splits * 2 <= kTile / VEC) splits *= 2;
  size_t lds = kGram2Bufs * ((size_t)(nb * kGram2B + (RG ? 3 : 0)) * (TB + 16) +
                             ((kTile + 15) & ~15));
  const size_t red = (size_t)kGram2Threads * kGram2B * kGram2B * sizeof(double);
  if (lds < red) lds = red;
  if (lds > 160 * 1024) return -2;
  const int64_t ntiles = (n + kTile - 1) / kTile;
  int64_t blocks = 256;                              // one 16-wave workgroup per CU
  if (blocks > kGramBlocks) blocks = kGramBlocks;
  if (blocks > ntiles) blocks = ntiles;
  auto kern = k_masked_gram_dma<T, TB, RG>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) { (void)hipGetLastError(); return -2; }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kGram2Threads), lds, st, P, nvec,
                     iwhere, n, nb, splits, ws, R);
  hipLaunchKernelGGL(k_gram2_final, dim3(nvec * (nvec + 1) / 2 + (RG ? nvec : 0)),
                     dim3(kBlock), 0, st, ws, (int)blocks, nvec, nb, result);
  return launch_status();
}

// rg: also form the reduced gradient (GramRG); tile size by the rows to stage
template <typename T>
int masked_gram_dma_launch(const GramPtrs<T> &P, int nvec, const int8_t *iwhere, int64_t n,
                           double *result, double *ws, hipStream_t st,
                           const GramRG<T> *rg = nullptr) {
  constexpr int VEC = 16 / (int)sizeof(T);
  if (n % VEC != 0 || (iwhere && n % 16 != 0) || nvec > kGram2MaxVec) return -2;
  const int rows = ((nvec + kGram2B - 1) / kGram2B) * kGram2B + (rg ? 3 : 0);
  if (rg) {
    for (int k = 0; k < 3; ++k)
      if (!rg->base[k] || ((uintptr_t)rg->base[k] & 15u)) return -2;
    if (rg->out && ((uintptr_t)rg->out & 15u)) return -2;
    if (g_gram_mfma) {
      const int rc = masked_gram_mfma_dispatch<T, true>(P, nvec, iwhere, n, result, ws, st, *rg);
      if (rc != -2) return rc;
    }
    if (rows <= 6) return masked_gram_dma_launch_tb<T, 8192, true>(P, nvec, iwhere, n, result, ws, st, *rg);
    if (rows <= 12) return masked_gram_dma_launch_tb<T, 4096, true>(P, nvec, iwhere, n, result, ws, st, *rg);
    return masked_gram_dma_launch_tb<T, 2048, true>(P, nvec, iwhere, n, result, ws, st, *rg);
  }
  const GramRG<T> none = GramRG<T>();
  if (g_gram_mfma) {
    const int rc = masked_gram_mfma_dispatch<T, false>(P, nvec, iwhere, n, result, ws, st, none);
    if (rc != -2) return rc;
  }
  if (rows <= 4) return masked_gram_dma_launch_tb<T, 8192, false>(P, nvec, iwhere, n, result, ws, st, none);
  if (rows <= 12) return masked_gram_dma_launch_tb<T, 4096, false>(P, nvec, iwhere, n, result, ws, st, none);
  return masked_gram_dma_launch_tb<T, 2048, false>(P, nvec, iwhere, n, result, ws, st, none);
}

template <typename T>
int masked_gram_impl(const T *const *vecs, int nvec, const int8_t *iwhere, int64_t n,
                     double *result, double *ws, void *stream,
                     const GramRG<T> *rg = nullptr) {
  if (!vecs || nvec < 1 || nvec > kGramMaxVec || n < 1 || !result || !ws)
    return NSOL_EINVAL;
  if (iwhere && ((uintptr_t)iwhere & 15u)) return NSOL_EINVAL;
  GramPtrs<T> P;
  for (int v = 0; v < kGramMax; ++v) {
    P.p[v] = v < nvec ? vecs[v] : nullptr;
    if (v < nvec && (!vecs[v] || ((uintptr_t)vecs[v] & 15u))) return NSOL_EINVAL;
  }
  if (g_gram_dma) {
    const int rc = masked_gram_dma_launch<T>(P, nvec, iwhere, n, result, ws,
                                             as_stream(stream), rg);
    if (rc != -2) return rc;
  }
  if (rg) return -2;             // (only the LDS-DMA staged kernel forms r as well)
  const int nb = (nvec + kGramB - 1) / kGramB;
  const int nblk = nb * (nb + 1) / 2;               // <= 36 for nvec <= 24
  int splits = 1;
  while (splits * 2 * nblk <= kBlock) splits *= 2;
  constexpr int VEC = 16 / sizeof(T);
  constexpr int kGramTile = gram_tile<T>();
  size_t lds = (size_t)nb * kGramB * (kGramTile + VEC) * sizeof(T);
  const size_t red = (size_t)kBlock * kGramB * kGramB * sizeof(double);
  if (lds < red) lds = red;
  const int64_t ntiles = (n + kGramTile - 1) / kGramTile;
  const int blocks = (int)(ntiles < kGramBlocks ? ntiles : kGramBlocks);
  auto kern = k_masked_gram<T>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(kBlock), lds, as_stream(stream), P, nvec,
                     iwhere, n, nb, splits, ws);
  hipLaunchKernelGGL(k_gram_final, dim3(nvec * (nvec + 1) / 2), dim3(kBlock), 0,
                     as_stream(stream), ws, blocks, nvec, nb, result);
  return launch_status();
}

// ---- Cauchy set-up: classify, d = -g on moving variables, breakpoints
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_cauchy_setup(
    const T *__restrict__ x, const T *__restrict__ g, int64_t n, T lo, T hi,
    bool has_lo, bool has_hi, int8_t *iw, T *__restrict__ d,
    T *__restrict__ tbk, double *ws) {
  typedef T V __attribute__((ext_vector_type(VEC)));
  typedef int8_t M __attribute__((ext_vector_type(VEC)));
  double a[4] = {0.0, 0.0, 0.0, 0.0};  // sum d^2, #breakpoints, #unbounded movers, #movers
  const T inf = (T)INFINITY;
  const int64_t nv = n / VEC;
  GRID_STRIDE(j, nv) {
    V xv, gv, dv, tv;
    M wv;
    if constexpr (VEC == 1) {
      xv[0] = x[j]; gv[0] = g[j]; wv[0] = iw[j];
    } else {
      xv = reinterpret_cast<const V *>(x)[j];
      gv = reinterpret_cast<const V *>(g)[j];
      wv = reinterpret_cast<const M *>(iw)[j];
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const T neg = -gv[e];
      int w = wv[e];
      const T tl = has_lo ? xv[e] - lo : inf;
      const T tu = has_hi ? hi - xv[e] : inf;
      if (w != 3 && w != -1) {
        const bool xlower = has_lo && tl <= T(0);
        const bool xupper = has_hi && tu <= T(0);
        w = 0;
        if (xlower) { if (neg <= T(0)) w = 1; }
        else if (xupper) { if (neg >= T(0)) w = 2; }
        else if (t_abs(neg) <= T(0)) w = -3;
        wv[e] = (int8_t)w;
      }
      T di = T(0), tb = inf;
      if (w == 0 || w == -1) {
        di = neg;
        a[0] += (double)neg * (double)neg;
        a[3] += 1.0;
        if (has_lo && neg < T(0)) { tb = tl / (-neg); a[1] += 1.0; }
        else if (has_hi && neg > T(0)) { tb = tu / neg; a[1] += 1.0; }
        else if (t_abs(neg) > T(0)) a[2] += 1.0;
      }
      dv[e] = di;
      tv[e] = tb;
    }
    if constexpr (VEC == 1) {
      iw[j] = wv[0]; d[j] = dv[0]; tbk[j] = tv[0];
    } else {
      reinterpret_cast<M *>(iw)[j] = wv;
      reinterpret_cast<V *>(d)[j] = dv;
      reinterpret_cast<V *>(tbk)[j] = tv;
    }
  }
  block_partials<4>(a, ws, false);
}

// ---- breakpoints (t, i) lexicographically after (t_done, i_done), t <= t_hi
// Compaction into out[] (any order; the candidates are sorted afterwards).  A
// workgroup looks at kSelSweeps x kSelPer elements per thread (16-byte loads, one
// bit of state per element) and reserves its range with ONE returning atomic:
// same-address atomics are what this kernel's time is made of -- one per 256
// elements cost 2.4 ms at 512^3, one per 4 096 elements 0.32 ms, one per wave
// and 1 024 elements (no barrier) was slower again; a plain scan of tbk takes
// 0.1 ms.
constexpr int kSelPer = 16;
constexpr int kSelSweeps = 8;

template <typename T>
__global__ __launch_bounds__(kBlock) void k_select(const T *__restrict__ tbk,
                                                    int64_t n, T t_done,
                                                    int64_t i_done, T t_hi,
                                                    int64_t *out, int capacity,
                                                    int *count) {
  __shared__ int s_cnt[kBlock / kWave];
  __shared__ int s_base;
  constexpr int VEC = 4;
  constexpr int NV = kSelPer / VEC;
  typedef T V __attribute__((ext_vector_type(VEC)));
  const int64_t chunk = (int64_t)kBlock * kSelPer;          // elements per sweep
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  const bool al = (reinterpret_cast<uintptr_t>(tbk) % sizeof(V)) == 0;
  const int64_t first = (int64_t)blockIdx.x * kSelSweeps * chunk;
  // thread t looks, in each of kSelSweeps sweeps, at NV groups of VEC consecutive
  // elements, groups kBlock apart; one bit per element
  unsigned flags[kSelSweeps];
  int mine = 0;
#pragma unroll
  for (int sw = 0; sw < kSelSweeps; ++sw) {
    const int64_t base = first + sw * chunk;
    unsigned f = 0;
#pragma unroll
    for (int r = 0; r < NV; ++r) {
      const int64_t i0 = base + ((int64_t)r * kBlock + threadIdx.x) * VEC;
      T t4[VEC];
      if (al && i0 + VEC <= n) {
        const V v = *reinterpret_cast<const V *>(tbk + i0);
#pragma unroll
        for (int k = 0; k < VEC; ++k) t4[k] = v[k];
      } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k) t4[k] = (i0 + k < n) ? tbk[i0 + k] : t_hi;
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const int64_t i = i0 + k;
        const T t = t4[k];
        if (i < n && t <= t_hi && (t > t_done || (t == t_done && i > i_done)))
          f |= 1u << (r * VEC + k);
      }
    }
    flags[sw] = f;
    mine += __popc(f);
  }
  // exclusive prefix of `mine` inside the wave, then over the waves
  int incl = mine;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const int up = __shfl_up(incl, d, kWave);
    if (lane >= d) incl += up;
  }
  if (lane == kWave - 1) s_cnt[wv] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int k = 0; k < kBlock / kWave; ++k) tot += s_cnt[k];
    s_base = tot ? atomicAdd(count, tot) : 0;
  }
  __syncthreads();
  if (mine) {
    int slot = s_base + incl - mine;
    for (int k = 0; k < wv; ++k) slot += s_cnt[k];
#pragma unroll
    for (int sw = 0; sw < kSelSweeps; ++sw) {
#pragma unroll
      for (int b = 0; b < kSelPer; ++b) {
        if (flags[sw] & (1u << b)) {
          const int64_t i = first + sw * chunk +
                            ((int64_t)(b / VEC) * kBlock + threadIdx.x) * VEC + b % VEC;
          if (slot < capacity) out[slot] = i;
          ++slot;
        }
      }
    }
  }
}

// how many breakpoints the window holds (no atomics: used to size the window)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_count_window(
    const T *__restrict__ tbk, int64_t n, T t_done, int64_t i_done, T t_hi,
    double *ws) {
  double a[1] = {0.0};
  GRID_STRIDE(i, n) {
    const T t = tbk[i];
    if (t <= t_hi && (t > t_done || (t == t_done && i > i_done))) a[0] += 1.0;
  }
  block_partials<1>(a, ws, false);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_gather(const T *__restrict__ src,
                                                    const int64_t *idx,
                                                    int count, T *out) {
  GRID_STRIDE(j, count) out[j] = src[idx[j]];
}

// VEC elements (16 bytes) per lane and trip where the arrays allow it; mask bytes are
// written only where a variable became fixed and never read.  (Measured at 512^3: 0.42 ms
// in either form -- 16 B per voxel at 0.63 of the HBM peak is what four streams reach
// here, the access width is not what bounds it; a form that also read the mask: 0.49.)
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_cauchy_finish(
    const T *__restrict__ x, const T *__restrict__ d,
    const T *__restrict__ tbk, int64_t n, T lo, T hi, int8_t *iw,
    T *__restrict__ xcp, T tsum, T t_done, int64_t i_done) {
  if constexpr (VEC == 1) {
    GRID_STRIDE(i, n) {
      const T t = tbk[i];
      const bool fixed = (t < t_done) || (t == t_done && i <= i_done);
      if (fixed) {
        const bool up = d[i] > T(0);
        xcp[i] = up ? hi : lo;
        iw[i] = up ? 2 : 1;
      } else {
        xcp[i] = x[i] + tsum * d[i];
      }
    }
  } else {
    typedef T V __attribute__((ext_vector_type(VEC)));
    const int64_t nv = n / VEC;
    GRID_STRIDE(j, nv) {
      const V tv = reinterpret_cast<const V *>(tbk)[j];
      const V xv = reinterpret_cast<const V *>(x)[j];
      const V dv = reinterpret_cast<const V *>(d)[j];
      V out;
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const int64_t i = j * VEC + k;
        const bool fixed = (tv[k] < t_done) || (tv[k] == t_done && i <= i_done);
        const bool up = dv[k] > T(0);
        out[k] = fixed ? (up ? hi : lo) : xv[k] + tsum * dv[k];
        if (fixed) iw[i] = up ? 2 : 1;          // (few: the mask is not read)
      }
      reinterpret_cast<V *>(xcp)[j] = out;
    }
  }
}

// ---- out = mask ? scale * (sum base + sum_j c_j W_j) : 0
constexpr int kMaxW = 40;
// A pass over ten to twenty-odd vectors runs faster with FEWER workgroups than the
// element-wise kernels' cap (fewer DRAM streams open at a time): the assembly of LSMR's
// solution from ten vectors at 512^3 takes 1.37 / 1.10 / 1.08 / 1.34 ms with 2 048 /
// 1 024 / 512 / 256 workgroups, the 23-vector sums of L-BFGS-B 2.6 / 2.5 / 2.5 ms
// (tools/_probe/wcomb_time.py, lb_grid.py).  Half the cap serves both.
inline int wcomb_grid(int64_t n) {
  const int g = grid_for(n), cap = g_max_grid_blocks / 2 > 0 ? g_max_grid_blocks / 2 : 1;
  return g < cap ? g : cap;
}
template <typename T>
struct WComb {
  const T *base[3];
  T bcoef[3];
  const T *w[kMaxW];
  T wcoef[kMaxW];
  int nbase, nw;
};

// VEC elements = 16 bytes per lane and trip; the W vectors are fetched four at a
// time so that several loads are in flight (with one 4-byte load after the other
// the kernel ran at 3.6 TB/s for ten stored pairs); the sum keeps its order.
// CLIP: the sum is projected onto [lo, hi] as nsol_clip_* does (LSMR's solution
// assembled from its stored vectors and clipped to the solver's bounds in one pass,
// tikhonov_linear_solver.py:142-158)
template <typename T, int VEC, bool CLIP = false>
__global__ __launch_bounds__(kBlock) void k_wcomb(T *__restrict__ out, int64_t n,
                                                   const int8_t *iw, T scale,
                                                   WComb<T> C, T lo = T(0), T hi = T(0)) {
  typedef T V __attribute__((ext_vector_type(VEC)));
  typedef int8_t M __attribute__((ext_vector_type(VEC)));
  const int64_t nv = n / VEC;
  GRID_STRIDE(j, nv) {
    M m;
    bool any = true;
    if (iw) {
      m = reinterpret_cast<const M *>(iw)[j];
      any = false;
#pragma unroll
      for (int e = 0; e < VEC; ++e) any = any || (m[e] <= 0);
    }
    V acc = V(T(0));
    if (any) {
      for (int k = 0; k < C.nbase; ++k)
        acc += C.bcoef[k] * reinterpret_cast<const V *>(C.base[k])[j];
      int q = 0;
      for (; q + 4 <= C.nw; q += 4) {
        const V a0 = reinterpret_cast<const V *>(C.w[q])[j];
        const V a1 = reinterpret_cast<const V *>(C.w[q + 1])[j];
        const V a2 = reinterpret_cast<const V *>(C.w[q + 2])[j];
        const V a3 = reinterpret_cast<const V *>(C.w[q + 3])[j];
        acc += C.wcoef[q] * a0;
        acc += C.wcoef[q + 1] * a1;
        acc += C.wcoef[q + 2] * a2;
        acc += C.wcoef[q + 3] * a3;
      }
      for (; q < C.nw; ++q) acc += C.wcoef[q] * reinterpret_cast<const V *>(C.w[q])[j];
      acc *= scale;
      if constexpr (CLIP) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          T v = acc[e];
          v = v < lo ? lo : v;
          acc[e] = v > hi ? hi : v;
        }
      }
      if (iw) {
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if (m[e] > 0) acc[e] = T(0);
      }
    }
    reinterpret_cast<V *>(out)[j] = acc;
  }
}

// ---- projected subspace step
template <typename T>
__global__ __launch_bounds__(kBlock) void k_project_step(
    const T *__restrict__ xcp, const T *__restrict__ d, int64_t n, T lo, T hi,
    bool has_lo, bool has_hi, const int8_t *iw, T *__restrict__ xn,
    double *ws) {
  double a[1] = {0.0};
  GRID_STRIDE(i, n) {
    T v = xcp[i];
    if (!iw || iw[i] <= 0) {
      v = v + d[i];
      if (has_lo && v < lo) v = lo;
      if (has_hi && v > hi) v = hi;
      if ((has_lo && v == lo) || (has_hi && v == hi)) a[0] += 1.0;
    }
    xn[i] = v;
  }
  block_partials<1>(a, ws, false);
}

// the feasible step ratio of one variable: d<0: (lo-x)/d (0 if lo-x >= 0); d>0: (hi-x)/d
template <typename T>
__device__ __forceinline__ double step_ratio(T xv, T dv, T lo, T hi, bool has_lo,
                                             bool has_hi) {
  if (dv < T(0) && has_lo) {
    const T t2 = lo - xv;
    return (t2 >= T(0)) ? 0.0 : (double)(t2 / dv);
  }
  if (dv > T(0) && has_hi) {
    const T t2 = hi - xv;
    return (t2 <= T(0)) ? 0.0 : (double)(t2 / dv);
  }
  return INFINITY;
}

// ---- min over (masked) i of the feasible step ratio, with the smallest index
//      among ties; ratios: d<0: (lo-x)/d (0 if lo-x >= 0); d>0: (hi-x)/d

template <typename T>
__global__ __launch_bounds__(kBlock) void k_ratio_min(
    const T *__restrict__ x, const T *__restrict__ d, int64_t n, T lo, T hi,
    bool has_lo, bool has_hi, const int8_t *iw, double *ws) {
  double best = INFINITY;
  int64_t bidx = -1;
  GRID_STRIDE(i, n) {
    if (!iw || iw[i] <= 0) {
      const double r = step_ratio(x[i], d[i], lo, hi, has_lo, has_hi);
      if (r < best) { best = r; bidx = i; }   // grid-stride: i increases
    }
  }
  __shared__ double sv[kBlock];
  __shared__ int64_t si[kBlock];
  sv[threadIdx.x] = best;
  si[threadIdx.x] = bidx;
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const double v2 = sv[threadIdx.x + s];
      const int64_t i2 = si[threadIdx.x + s];
      const double v1 = sv[threadIdx.x];
      const int64_t i1 = si[threadIdx.x];
      if (v2 < v1 || (v2 == v1 && i2 >= 0 && (i1 < 0 || i2 < i1))) {
        sv[threadIdx.x] = v2;
        si[threadIdx.x] = i2;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    ws[blockIdx.x] = sv[0];
    ws[kRed + blockIdx.x] = (double)si[0];
  }
}

// (min, smallest index among ties) over the blocks' partials: 64 lanes stride over
// them (one thread walking all 1024 took 0.16 ms), then a shuffle reduction with the
// same ordering rule
__global__ void k_ratio_final(const double *ws, int nparts, double *result) {
  double v = INFINITY, idx = -1.0;
  for (int j = threadIdx.x; j < nparts; j += kWave) {
    const double v2 = ws[j], i2 = ws[kRed + j];
    if (v2 < v || (v2 == v && i2 >= 0 && (idx < 0 || i2 < idx))) {
      v = v2;
      idx = i2;
    }
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
    const double v2 = __shfl_down(v, off, kWave), i2 = __shfl_down(idx, off, kWave);
    if (v2 < v || (v2 == v && i2 >= 0 && (idx < 0 || i2 < idx))) {
      v = v2;
      idx = i2;
    }
  }
  if (threadIdx.x == 0) {
    result[0] = v;
    result[1] = idx;
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_trunc_apply(
    const T *__restrict__ xcp, const T *__restrict__ d, int64_t n, T lo, T hi,
    const int8_t *iw, T alpha, int64_t ibd, T *__restrict__ xn) {
  GRID_STRIDE(i, n) {
    T v = xcp[i];
    if (!iw || iw[i] <= 0) {
      if (i == ibd) v = (d[i] > T(0)) ? hi : lo;
      else v = v + alpha * d[i];
    }
    xn[i] = v;
  }
}

template <typename T>
inline T cast_bound(double b) {
  if (b == INFINITY) return (T)INFINITY;
  if (b == -INFINITY) return (T)(-INFINITY);
  return (T)b;
}

// ---- the subspace step in ONE pass over the stored vectors ---------------------
// After the solve with the subspace matrix an L-BFGS-B iteration forms, one after the
// other (scipy's subsm, lnsrlb, matupd behind tikhonov_linear_solver.py:214-220):
//   dsub = free ? scale * (r + sum_j c_j W_j) : 0                  (k_wcomb)
//   xn   = free ? clip(xcp + dsub) : xcp,  hits = #{xn at a bound}  (k_project_step)
//   d    = xn - x,  d'd,  g'd                                       (k_diff_dots)
//   W_j' d for every stored vector (the new row of S'S and S'Y)     (k_mdots)
//   the largest feasible step along d from x                        (k_ratio_min)
// -- four passes, two of them over all 2c stored vectors.  Here a lane keeps its 16
// bytes of every W_j in registers: the combination, the projection, the difference and
// the 2c + 3 sums come from one read of W (105 bytes per voxel at ten stored pairs
// instead of 215).  The arithmetic per value is that of the four kernels, in the same
// order; the sums are accumulated per lane in the order k_diff_dots / k_mdots use.
// RIN: r is not read but formed here, r = free ? rb[0] xcp + rb[1] x + rb[2] g + sum_j
// rcoef[j] W_j : 0 -- nsol_lb_wcomb_*'s sum (scipy's cmprlb) term for term in its order,
// from the values the step holds anyway (the Gram pass then neither forms nor stores r,
// and this pass does not read it).
template <typename T>
struct SubStep {
  const T *w[kDotsMax];
  T wcoef[kDotsMax];
  int nw;
};
template <typename T>
struct SubStepR {
  T rb[3];
  T rcoef[kDotsMax];
};

template <typename T, int VEC, int NV, bool RIN = false>
__global__ __launch_bounds__(kBlock) void k_subspace_step(
    SubStep<T> C, const T *__restrict__ r, const T *__restrict__ xcp,
    const T *__restrict__ x, const T *__restrict__ g, const int8_t *iw, int64_t n,
    T scale, T lo, T hi, bool has_lo, bool has_hi, T *__restrict__ xn_out,
    T *__restrict__ d_out, double *ws, SubStepR<T> RC = SubStepR<T>()) {
  typedef T V __attribute__((ext_vector_type(VEC)));
  typedef int8_t M __attribute__((ext_vector_type(VEC)));
  double a[NV + 3];                          // hits, d'd, g'd, W_j'd
#pragma unroll
  for (int k = 0; k < NV + 3; ++k) a[k] = 0.0;
  double ratio = INFINITY;                   // min over all variables (lnsrlb's stpmx)
  const int64_t nv = n / VEC;
  GRID_STRIDE(j, nv) {
    V wv[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k)
      wv[k] = k < C.nw ? reinterpret_cast<const V *>(C.w[k])[j] : V(T(0));
    const V cv = reinterpret_cast<const V *>(xcp)[j];
    const V xv = reinterpret_cast<const V *>(x)[j];
    const V gv = reinterpret_cast<const V *>(g)[j];
    M m = M((int8_t)0);
    if (iw) m = reinterpret_cast<const M *>(iw)[j];
    V rv;
    if constexpr (RIN) {
      rv = V(T(0));
      rv += RC.rb[0] * cv;
      rv += RC.rb[1] * xv;
      rv += RC.rb[2] * gv;
#pragma unroll
      for (int k = 0; k < NV; ++k)
        if (k < C.nw) rv += RC.rcoef[k] * wv[k];
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        if (m[e] > 0) rv[e] = T(0);
    } else {
      rv = reinterpret_cast<const V *>(r)[j];
    }
    V acc = V(T(0));
    acc += T(1) * rv;
#pragma unroll
    for (int k = 0; k < NV; ++k)
      if (k < C.nw) acc += C.wcoef[k] * wv[k];
    acc *= scale;
    V xn, dv;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      T v = cv[e];
      if (m[e] <= 0) {
        v = v + acc[e];
        if (has_lo && v < lo) v = lo;
        if (has_hi && v > hi) v = hi;
        if ((has_lo && v == lo) || (has_hi && v == hi)) a[0] += 1.0;
      }
      xn[e] = v;
      dv[e] = T(1) * v + T(-1) * xv[e];
      ratio = fmin(ratio, step_ratio(xv[e], dv[e], lo, hi, has_lo, has_hi));
    }
    reinterpret_cast<V *>(xn_out)[j] = xn;
    reinterpret_cast<V *>(d_out)[j] = dv;
#pragma unroll
    for (int e = 0; e < VEC; ++e) a[1] += (double)dv[e] * (double)dv[e];
#pragma unroll
    for (int e = 0; e < VEC; ++e) a[2] += (double)dv[e] * (double)gv[e];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      if (k < C.nw) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) a[3 + k] += (double)wv[k][e] * (double)dv[e];
      }
    }
  }
  block_partials<NV + 3>(a, ws, false);
  double neg[1] = {-ratio};
  block_partials<1>(neg, ws + (int64_t)(NV + 3) * kRed, true);
}

// returns -2 (nothing launched) where the vector form does not apply
template <typename T>
int subspace_step_impl(const T *const *w_host, const double *wcoef_host, int nw,
                       const T *r, const T *xcp, const T *x, const T *g,
                       const int8_t *iwhere, int64_t n, double scale, double lo, double hi,
                       T *xn, T *d, double *result, double *ws, void *stream,
                       const double *rb3_host = nullptr, const double *rcoef_host = nullptr) {
  // r == NULL: formed in the kernel from rb3_host / rcoef_host (both then required)
  if (n < 1 || nw < 1 || !w_host || !wcoef_host || (!r && (!rb3_host || !rcoef_host)) ||
      !xcp || !x || !g || !xn || !d || !result || !ws)
    return NSOL_EINVAL;
  if (nw > kDotsMax) return -2;
  constexpr int VW = 16 / sizeof(T);
  uintptr_t bits = reinterpret_cast<uintptr_t>(r) | reinterpret_cast<uintptr_t>(xcp) |
                   reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(g) |
                   reinterpret_cast<uintptr_t>(xn) | reinterpret_cast<uintptr_t>(d);
  SubStep<T> C;
  C.nw = nw;
  for (int k = 0; k < kDotsMax; ++k) {
    C.w[k] = k < nw ? w_host[k] : nullptr;
    C.wcoef[k] = k < nw ? (T)wcoef_host[k] : T(0);
    if (k < nw) {
      if (!w_host[k]) return NSOL_EINVAL;
      bits |= reinterpret_cast<uintptr_t>(w_host[k]);
    }
  }
  if (n % VW != 0 || (bits & 15) ||
      (iwhere && (reinterpret_cast<uintptr_t>(iwhere) & (VW - 1))))
    return -2;
  const int gr = rgrid(n / VW);
  const T tlo = cast_bound<T>(lo), thi = cast_bound<T>(hi);
  SubStepR<T> RC;
  for (int k = 0; k < 3; ++k) RC.rb[k] = (!r) ? (T)rb3_host[k] : T(0);
  for (int k = 0; k < kDotsMax; ++k) RC.rcoef[k] = (!r && k < nw) ? (T)rcoef_host[k] : T(0);
  if (!r) {
    if (nw > 12)
      hipLaunchKernelGGL((k_subspace_step<T, VW, kDotsMax, true>), dim3(gr), dim3(kBlock), 0,
                         as_stream(stream), C, r, xcp, x, g, iwhere, n, (T)scale, tlo, thi,
                         lo > -INFINITY, hi < INFINITY, xn, d, ws, RC);
    else
      hipLaunchKernelGGL((k_subspace_step<T, VW, 12, true>), dim3(gr), dim3(kBlock), 0,
                         as_stream(stream), C, r, xcp, x, g, iwhere, n, (T)scale, tlo, thi,
                         lo > -INFINITY, hi < INFINITY, xn, d, ws, RC);
  } else if (nw > 12)
    hipLaunchKernelGGL((k_subspace_step<T, VW, kDotsMax>), dim3(gr), dim3(kBlock), 0,
                       as_stream(stream), C, r, xcp, x, g, iwhere, n, (T)scale, tlo, thi,
                       lo > -INFINITY, hi < INFINITY, xn, d, ws, RC);
  else
    hipLaunchKernelGGL((k_subspace_step<T, VW, 12>), dim3(gr), dim3(kBlock), 0,
                       as_stream(stream), C, r, xcp, x, g, iwhere, n, (T)scale, tlo, thi,
                       lo > -INFINITY, hi < INFINITY, xn, d, ws, RC);
  hipLaunchKernelGGL(k_final, dim3(1), dim3(kBlock), 0, as_stream(stream), ws, gr, nw + 3,
                     false, result);
  // result[nw + 3] = -(smallest feasible step ratio), -inf when no variable limits it
  hipLaunchKernelGGL(k_final, dim3(1), dim3(kBlock), 0, as_stream(stream),
                     ws + (int64_t)((nw > 12 ? kDotsMax : 12) + 3) * kRed, gr, 1, true,
                     result + nw + 3);
  return launch_status();
}

template <typename T>
int gram_rgrad(const T *const *vecs, int nvec, const int8_t *iwhere, int64_t n,
               double *result, double *ws, const T *const *base3,
               const double *bcoef3, const double *wcoef, T *r_out, void *stream) {
  if (!base3 || !bcoef3 || !wcoef || nvec > kGram2MaxVec) return NSOL_EINVAL;
  GramRG<T> R;
  for (int k = 0; k < 3; ++k) { R.base[k] = base3[k]; R.bcoef[k] = (T)bcoef3[k]; }
  for (int k = 0; k < kGram2MaxVec; ++k) R.wcoef[k] = k < nvec ? (T)wcoef[k] : T(0);
  R.out = r_out;
  return masked_gram_impl<T>(vecs, nvec, iwhere, n, result, ws, stream, &R);
}

}  // namespace

extern "C" {

int nsol_lb_count_free(const int8_t *iwhere, int64_t n, double *result,
                       double *ws, void *stream) {
  if (n < 0 || !result || !ws || (n > 0 && !iwhere)) return NSOL_EINVAL;
  const bool vec = n % 16 == 0 && !(reinterpret_cast<uintptr_t>(iwhere) & 15);
  const int g = rgrid(vec ? n / 16 : n);
  if (vec)
    hipLaunchKernelGGL(k_count_free<16>, dim3(g), dim3(kBlock), 0, as_stream(stream),
                       iwhere, n, ws);
  else
    hipLaunchKernelGGL(k_count_free<1>, dim3(g), dim3(kBlock), 0, as_stream(stream),
                       iwhere, n, ws);
  hipLaunchKernelGGL(k_final, dim3(1), dim3(kBlock), 0, as_stream(stream), ws, g,
                     1, false, result);
  return launch_status();
}

extern "C" {
/* experiment knobs of this file: "lb_gram_dma", "lb_gram_mfma" */
int nsol_hip_set_param_lb(const char *name, int value) {
  if (!name) return NSOL_EINVAL;
  if (!strcmp(name, "lb_gram_dma")) g_gram_dma = value;
  else if (!strcmp(name, "lb_gram_mfma")) g_gram_mfma = value;
  else return NSOL_EINVAL;
  return 0;
}
int64_t nsol_lb_gram_ws_doubles(void) {
  return (int64_t)kGramBlocks * (kGramEnt > kGram2Ent ? kGramEnt : kGram2Ent);
}
int nsol_lb_mdots_f32(const float *const *vecs, int nvec, const float *y,
                      const int8_t *iwhere, int64_t n, double *result, double *ws,
                      void *stream) {
  return mdots_impl<float>(vecs, nvec, y, iwhere, n, result, ws, stream);
}
int nsol_lb_mdots_f64(const double *const *vecs, int nvec, const double *y,
                      const int8_t *iwhere, int64_t n, double *result, double *ws,
                      void *stream) {
  return mdots_impl<double>(vecs, nvec, y, iwhere, n, result, ws, stream);
}
int nsol_lb_diff_dots_f32(const float *a, const float *b, const float *c, float *out,
                          int64_t n, double *result, double *ws, void *stream) {
  return diff_dots_impl<float>(a, b, c, out, n, result, ws, stream);
}
int nsol_lb_diff_dots_f64(const double *a, const double *b, const double *c,
                          double *out, int64_t n, double *result, double *ws,
                          void *stream) {
  return diff_dots_impl<double>(a, b, c, out, n, result, ws, stream);
}
int nsol_lb_masked_gram_f32(const float *const *vecs, int nvec, const int8_t *iwhere,
                            int64_t n, double *result, double *ws, void *stream) {
  return masked_gram_impl<float>(vecs, nvec, iwhere, n, result, ws, stream);
}
int nsol_lb_masked_gram_f64(const double *const *vecs, int nvec, const int8_t *iwhere,
                            int64_t n, double *result, double *ws, void *stream) {
  return masked_gram_impl<double>(vecs, nvec, iwhere, n, result, ws, stream);
}
}

extern "C" {
int nsol_lb_subspace_step_f32(const float *const *w_host, const double *wcoef_host, int nw,
                              const float *r, const float *xcp, const float *x,
                              const float *g, const int8_t *iwhere, int64_t n,
                              double scale, double lo, double hi, float *xn, float *d,
                              double *result, double *ws, void *stream) {
  return subspace_step_impl<float>(w_host, wcoef_host, nw, r, xcp, x, g, iwhere, n, scale,
                                   lo, hi, xn, d, result, ws, stream);
}
int nsol_lb_subspace_step_f64(const double *const *w_host, const double *wcoef_host, int nw,
                              const double *r, const double *xcp, const double *x,
                              const double *g, const int8_t *iwhere, int64_t n,
                              double scale, double lo, double hi, double *xn, double *d,
                              double *result, double *ws, void *stream) {
  return subspace_step_impl<double>(w_host, wcoef_host, nw, r, xcp, x, g, iwhere, n, scale,
                                    lo, hi, xn, d, result, ws, stream);
}
int nsol_lb_subspace_step_r_f32(const float *const *w_host, const double *wcoef_host, int nw,
                                const double *rb3_host, const double *rcoef_host,
                                const float *xcp, const float *x, const float *g,
                                const int8_t *iwhere, int64_t n, double scale, double lo,
                                double hi, float *xn, float *d, double *result, double *ws,
                                void *stream) {
  if (!rb3_host || !rcoef_host) return NSOL_EINVAL;
  return subspace_step_impl<float>(w_host, wcoef_host, nw, nullptr, xcp, x, g, iwhere, n,
                                   scale, lo, hi, xn, d, result, ws, stream, rb3_host,
                                   rcoef_host);
}
int nsol_lb_subspace_step_r_f64(const double *const *w_host, const double *wcoef_host, int nw,
                                const double *rb3_host, const double *rcoef_host,
                                const double *xcp, const double *x, const double *g,
                                const int8_t *iwhere, int64_t n, double scale, double lo,
                                double hi, double *xn, double *d, double *result, double *ws,
                                void *stream) {
  if (!rb3_host || !rcoef_host) return NSOL_EINVAL;
  return subspace_step_impl<double>(w_host, wcoef_host, nw, nullptr, xcp, x, g, iwhere, n,
                                    scale, lo, hi, xn, d, result, ws, stream, rb3_host,
                                    rcoef_host);
}
int nsol_lb_masked_gram_rgrad_f32(const float *const *vecs, int nvec,
                                  const int8_t *iwhere, int64_t n, double *result,
                                  double *ws, const float *const *base3,
                                  const double *bcoef3, const double *wcoef,
                                  float *r_out, void *stream) {
  return gram_rgrad<float>(vecs, nvec, iwhere, n, result, ws, base3, bcoef3, wcoef, r_out,
                           stream);
}
int nsol_lb_masked_gram_rgrad_f64(const double *const *vecs, int nvec,
                                  const int8_t *iwhere, int64_t n, double *result,
                                  double *ws, const double *const *base3,
                                  const double *bcoef3, const double *wcoef,
                                  double *r_out, void *stream) {
  return gram_rgrad<double>(vecs, nvec, iwhere, n, result, ws, base3, bcoef3, wcoef, r_out,
                            stream);
}
}

#define NSOL_LB_DEF(T, SUF)                                                      \
  int nsol_lb_projgr_##SUF(const T *x, const T *g, int64_t n, double lo,         \
                           double hi, double *result, double *ws, void *s) {     \
    if (n < 1 || !x || !g || !result || !ws) return NSOL_EINVAL;                 \
    constexpr int VW = 16 / sizeof(T);                                           \
    const bool vec = n % VW == 0 && !((reinterpret_cast<uintptr_t>(x) |          \
                                       reinterpret_cast<uintptr_t>(g)) & 15);    \
    const int gr = rgrid(vec ? n / VW : n);                                      \
    if (vec)                                                                     \
      hipLaunchKernelGGL((k_projgr<T, VW>), dim3(gr), dim3(kBlock), 0,           \
                         as_stream(s), x, g, n, cast_bound<T>(lo),               \
                         cast_bound<T>(hi), lo > -INFINITY, hi < INFINITY, ws);  \
    else                                                                         \
      hipLaunchKernelGGL((k_projgr<T, 1>), dim3(gr), dim3(kBlock), 0,            \
                         as_stream(s), x, g, n, cast_bound<T>(lo),               \
                         cast_bound<T>(hi), lo > -INFINITY, hi < INFINITY, ws);  \
    hipLaunchKernelGGL(k_final, dim3(1), dim3(kBlock), 0, as_stream(s), ws, gr,  \
                       1, true, result);                                         \
    return launch_status();                                                      \
  }                                                                              \
  int nsol_lb_mdot_##SUF(const T *x, const T *y, const int8_t *iwhere,           \
                         int64_t n, double *result, double *ws, void *s) {       \
    if (n < 1 || !x || !y || !result || !ws) return NSOL_EINVAL;                 \
    constexpr int VW = 16 / sizeof(T);                                           \
    const bool vec = n % VW == 0 && !((uintptr_t)x & 15) && !((uintptr_t)y & 15) && \
                     (!iwhere || !((uintptr_t)iwhere & (VW - 1)));               \
    const int gr = rgrid(vec ? n / VW : n);                                      \
    if (vec)                                                                     \
      hipLaunchKernelGGL((k_mdot<T, VW>), dim3(gr), dim3(kBlock), 0,             \
                         as_stream(s), x, y, iwhere, n, ws);                     \
    else                                                                         \
      hipLaunchKernelGGL((k_mdot<T, 1>), dim3(gr), dim3(kBlock), 0,              \
                         as_stream(s), x, y, iwhere, n, ws);                     \
    hipLaunchKernelGGL(k_final, dim3(1), dim3(kBlock), 0, as_stream(s), ws, gr,  \
                       1, false, result);                                        \
    return launch_status();                                                      \
  }                                                                              \
  int nsol_lb_cauchy_setup_##SUF(const T *x, const T *g, int64_t n, double lo,   \
                                 double hi, int8_t *iwhere, T *d, T *tbk,        \
                                 double *result, double *ws, void *s) {          \
    if (n < 1 || !x || !g || !iwhere || !d || !tbk || !result || !ws)            \
      return NSOL_EINVAL;                                                        \
    constexpr int VW = 16 / sizeof(T);                                           \
    const bool vec = n % VW == 0 &&                                              \
        !((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(g) |     \
           reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(tbk)) & 15) && \
        !(reinterpret_cast<uintptr_t>(iwhere) & (VW - 1));                       \
    const int gr = rgrid(vec ? n / VW : n);                                      \
    if (vec)                                                                     \
      hipLaunchKernelGGL((k_cauchy_setup<T, VW>), dim3(gr), dim3(kBlock), 0,     \
                         as_stream(s), x, g, n, cast_bound<T>(lo),               \
                         cast_bound<T>(hi), lo > -INFINITY, hi < INFINITY,       \
                         iwhere, d, tbk, ws);                                    \
    else                                                                         \
      hipLaunchKernelGGL((k_cauchy_setup<T, 1>), dim3(gr), dim3(kBlock), 0,      \
                         as_stream(s), x, g, n, cast_bound<T>(lo),               \
                         cast_bound<T>(hi), lo > -INFINITY, hi < INFINITY,       \
                         iwhere, d, tbk, ws);                                    \
    hipLaunchKernelGGL(k_final, dim3(1), dim3(kBlock), 0, as_stream(s), ws, gr,  \
                       4, false, result);                                        \
    return launch_status();                                                      \
  }                                                                              \
  int nsol_lb_select_##SUF(const T *tbk, int64_t n, double t_done,               \
                           int64_t i_done, double t_hi, int64_t *out_idx,        \
                           int capacity, int *count, void *s) {                  \
    if (n < 1 || !tbk || !out_idx || !count || capacity < 1)                     \
      return NSOL_EINVAL;                                                        \
    hipError_t e = hipMemsetAsync(count, 0, sizeof(int), as_stream(s));          \
    if (e != hipSuccess) return (int)e;                                          \
    const int64_t per = (int64_t)kBlock * kSelPer * kSelSweeps;                  \
    const int64_t blocks = (n + per - 1) / per;                                  \
    if (blocks > 0x7fffffff) return NSOL_EINVAL;                                 \
    hipLaunchKernelGGL(k_select<T>, dim3((unsigned)blocks), dim3(kBlock), 0,     \
                       as_stream(s), tbk, n, (T)t_done, i_done, (T)t_hi,         \
                       out_idx, capacity, count);                                \
    return launch_status();                                                      \
  }                                                                              \
  int nsol_lb_count_window_##SUF(const T *tbk, int64_t n, double t_done,         \
                                 int64_t i_done, double t_hi, double *result,    \
                                 double *ws, void *s) {                          \
    if (n < 1 || !tbk || !result || !ws) return NSOL_EINVAL;                     \
    const int gr = rgrid(n);                                                     \
    hipLaunchKernelGGL(k_count_window<T>, dim3(gr), dim3(kBlock), 0,             \
                       as_stream(s), tbk, n, (T)t_done, i_done, (T)t_hi, ws);    \
    hipLaunchKernelGGL(k_final, dim3(1), dim3(kBlock), 0, as_stream(s), ws, gr,  \
                       1, false, result);                                        \
    return launch_status();                                                      \
  }                                                                              \
  int nsol_lb_gather_##SUF(const T *src, const int64_t *idx, int count, T *out,  \
                           void *s) {                                            \
    if (count < 0 || !src || !idx || !out) return NSOL_EINVAL;                   \
    if (count == 0) return 0;                                                    \
    hipLaunchKernelGGL(k_gather<T>, dim3(grid_for(count)), dim3(kBlock), 0,      \
                       as_stream(s), src, idx, count, out);                      \
    return launch_status();                                                      \
  }                                                                              \
  int nsol_lb_cauchy_finish_##SUF(const T *x, const T *d, const T *tbk,          \
                                  int64_t n, double lo, double hi,               \
                                  int8_t *iwhere, T *xcp, double tsum,           \
                                  double t_done, int64_t i_done, void *s) {      \
    if (n < 1 || !x || !d || !tbk || !iwhere || !xcp) return NSOL_EINVAL;        \
    constexpr int VW = 16 / sizeof(T);                                           \
    const bool vec = n % VW == 0 &&                                              \
        !((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(d) |     \
           reinterpret_cast<uintptr_t>(tbk) | reinterpret_cast<uintptr_t>(xcp)) & 15) && \
        !(reinterpret_cast<uintptr_t>(iwhere) & (VW - 1));                       \
    if (vec)                                                                     \
      hipLaunchKernelGGL((k_cauchy_finish<T, VW>), dim3(grid_for(n / VW)),       \
                         dim3(kBlock), 0, as_stream(s), x, d, tbk, n,            \
                         cast_bound<T>(lo), cast_bound<T>(hi), iwhere, xcp,      \
                         (T)tsum, (T)t_done, i_done);                            \
    else                                                                         \
      hipLaunchKernelGGL((k_cauchy_finish<T, 1>), dim3(grid_for(n)), dim3(kBlock), \
                         0, as_stream(s), x, d, tbk, n, cast_bound<T>(lo),       \
                         cast_bound<T>(hi), iwhere, xcp, (T)tsum, (T)t_done,     \
                         i_done);                                                \
    return launch_status();                                                      \
  }                                                                              \
  int nsol_lb_wcomb_##SUF(T *out, int64_t n, const int8_t *iwhere, double scale, \
                          int nbase, const T *const *base_host,                  \
                          const double *bcoef_host, int nw,                      \
                          const T *const *w_host, const double *wcoef_host,      \
                          void *s) {                                             \
    if (n < 1 || !out || nbase < 0 || nbase > 3 || nw < 0 || nw > kMaxW)         \
      return NSOL_EINVAL;                                                        \
    WComb<T> C;                                                                  \
    C.nbase = nbase;                                                             \
    C.nw = nw;                                                                   \
    for (int k = 0; k < 3; ++k) {                                                \
      C.base[k] = k < nbase ? base_host[k] : nullptr;                            \
      C.bcoef[k] = k < nbase ? (T)bcoef_host[k] : T(0);                          \
    }                                                                            \
    for (int j = 0; j < kMaxW; ++j) {                                            \
      C.w[j] = j < nw ? w_host[j] : nullptr;                                     \
      C.wcoef[j] = j < nw ? (T)wcoef_host[j] : T(0);                             \
    }                                                                            \
    constexpr int VW = 16 / sizeof(T);                                           \
    bool vec = n % VW == 0 && !(reinterpret_cast<uintptr_t>(out) & 15) &&        \
               (!iwhere || !(reinterpret_cast<uintptr_t>(iwhere) & (VW - 1)));   \
    for (int k = 0; k < nbase; ++k)                                              \
      vec = vec && !(reinterpret_cast<uintptr_t>(base_host[k]) & 15);            \
    for (int j = 0; j < nw; ++j)                                                 \
      vec = vec && !(reinterpret_cast<uintptr_t>(w_host[j]) & 15);               \
    if (vec)                                                                     \
      hipLaunchKernelGGL((k_wcomb<T, VW>), dim3(wcomb_grid(n / VW)), dim3(kBlock), \
                         0, as_stream(s), out, n, iwhere, (T)scale, C);          \
    else                                                                         \
      hipLaunchKernelGGL((k_wcomb<T, 1>), dim3(wcomb_grid(n)), dim3(kBlock), 0,    \
                         as_stream(s), out, n, iwhere, (T)scale, C);             \
    return launch_status();                                                      \
  }                                                                              \
  int nsol_lincomb_clip_##SUF(T *out, int64_t n, int nw, const T *const *w_host, \
                              const double *wcoef_host, double lo, double hi,    \
                              void *s) {                                         \
    if (n < 1 || !out || nw < 1 || nw > kMaxW || !w_host || !wcoef_host ||       \
        !(lo <= hi))                                                             \
      return NSOL_EINVAL;                                                        \
    WComb<T> C;                                                                  \
    C.nbase = 0;                                                                 \
    C.nw = nw;                                                                   \
    for (int k = 0; k < 3; ++k) { C.base[k] = nullptr; C.bcoef[k] = T(0); }      \
    for (int j = 0; j < kMaxW; ++j) {                                            \
      C.w[j] = j < nw ? w_host[j] : nullptr;                                     \
      C.wcoef[j] = j < nw ? (T)wcoef_host[j] : T(0);                             \
      if (j < nw && !w_host[j]) return NSOL_EINVAL;                              \
    }                                                                            \
    constexpr int VW = 16 / sizeof(T);                                           \
    bool vec = n % VW == 0 && !(reinterpret_cast<uintptr_t>(out) & 15);          \
    for (int j = 0; j < nw; ++j)                                                 \
      vec = vec && !(reinterpret_cast<uintptr_t>(w_host[j]) & 15);               \
    if (vec)                                                                     \
      hipLaunchKernelGGL((k_wcomb<T, VW, true>), dim3(wcomb_grid(n / VW)),         \
                         dim3(kBlock), 0, as_stream(s), out, n,                  \
                         (const int8_t *)nullptr, T(1), C, cast_bound<T>(lo),    \
                         cast_bound<T>(hi));                                     \
    else                                                                         \
      hipLaunchKernelGGL((k_wcomb<T, 1, true>), dim3(wcomb_grid(n)), dim3(kBlock), \
                         0, as_stream(s), out, n, (const int8_t *)nullptr, T(1), \
                         C, cast_bound<T>(lo), cast_bound<T>(hi));               \
    return launch_status();                                                      \
  }                                                                              \
  int nsol_lb_project_step_##SUF(const T *xcp, const T *d, int64_t n, double lo, \
                                 double hi, const int8_t *iwhere, T *xnew,       \
                                 double *result, double *ws, void *s) {          \
    if (n < 1 || !xcp || !d || !xnew || !result || !ws) return NSOL_EINVAL;      \
    const int gr = rgrid(n);                                                     \
    hipLaunchKernelGGL(k_project_step<T>, dim3(gr), dim3(kBlock), 0,             \
                       as_stream(s), xcp, d, n, cast_bound<T>(lo),               \
                       cast_bound<T>(hi), lo > -INFINITY, hi < INFINITY, iwhere, \
                       xnew, ws);                                                \
    hipLaunchKernelGGL(k_final, dim3(1), dim3(kBlock), 0, as_stream(s), ws, gr,  \
                       1, false, result);                                        \
    return launch_status();                                                      \
  }                                                                              \
  int nsol_lb_ratio_min_##SUF(const T *x, const T *d, int64_t n, double lo,      \
                              double hi, const int8_t *iwhere, double *result,   \
                              double *ws, void *s) {                             \
    if (n < 1 || !x || !d || !result || !ws) return NSOL_EINVAL;                 \
    const int gr = rgrid(n);                                                     \
    hipLaunchKernelGGL(k_ratio_min<T>, dim3(gr), dim3(kBlock), 0, as_stream(s),  \
                       x, d, n, cast_bound<T>(lo), cast_bound<T>(hi),            \
                       lo > -INFINITY, hi < INFINITY, iwhere, ws);               \
    hipLaunchKernelGGL(k_ratio_final, dim3(1), dim3(64), 0, as_stream(s), ws,    \
                       gr, result);                                              \
    return launch_status();                                                      \
  }                                                                              \
  int nsol_lb_trunc_apply_##SUF(const T *xcp, const T *d, int64_t n, double lo,  \
                                double hi, const int8_t *iwhere, double alpha,   \
                                int64_t ibd, T *xnew, void *s) {                 \
    if (n < 1 || !xcp || !d || !xnew) return NSOL_EINVAL;                        \
    hipLaunchKernelGGL(k_trunc_apply<T>, dim3(grid_for(n)), dim3(kBlock), 0,     \
                       as_stream(s), xcp, d, n, cast_bound<T>(lo),               \
                       cast_bound<T>(hi), iwhere, (T)alpha, ibd, xnew);          \
    return launch_status();                                                      \
  }

NSOL_LB_DEF(float, f32)
NSOL_LB_DEF(double, f64)

}  // extern "C"
