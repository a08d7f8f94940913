// Thread <-> voxel mapping shared by the stand-alone stencil kernels.
//
// A 256-thread workgroup covers ROWS consecutive y-rows (one wave each when
// ROWS = 4, all four waves on one row when ROWS = 1) and 64*VEC*(4/ROWS)
// consecutive x per row; blockIdx.y walks y, blockIdx.z walks z.  No integer
// division is needed to recover (x, y, z), every own-voxel access is one
// 16-byte load/store when VEC > 1, and the +-1 neighbours along y / z are again
// aligned vectors (L2 hits: the adjacent row is read by the adjacent wave).
#pragma once

#include "nsol_common.hpp"

namespace nsol {

template <typename T, int V>
struct VecT {
  typedef T type __attribute__((ext_vector_type(V)));
};

template <typename T, int V>
__device__ __forceinline__ void vload(const T *p, T (&v)[V]) {
  if constexpr (V == 1) {
    v[0] = *p;
  } else {
    typedef typename VecT<T, V>::type P;
    const P t = *reinterpret_cast<const P *>(p);
#pragma unroll
    for (int k = 0; k < V; ++k) v[k] = t[k];
  }
}

template <typename T, int V>
__device__ __forceinline__ void vstore(T *p, const T (&v)[V]) {
  if constexpr (V == 1) {
    *p = v[0];
  } else {
    typedef typename VecT<T, V>::type P;
    P t;
#pragma unroll
    for (int k = 0; k < V; ++k) t[k] = v[k];
    *reinterpret_cast<P *>(p) = t;
  }
}

template <typename T, int V>
__device__ __forceinline__ void vzero(T (&v)[V]) {
#pragma unroll
  for (int k = 0; k < V; ++k) v[k] = T(0);
}

struct Voxel {
  int64_t ix, iy, iz, i;
  bool ok;
  int nval;   // elements of the lane's vector that lie inside the row (VEC unless ragged)
  bool wide;  // a whole vector may be READ at this voxel and at its forward neighbours
              // (next row / plane) even when nval < VEC: the bytes behind the row's end
              // belong to the same array
};

// Rows that are not a multiple of the vector width (RAG kernels): vectors start
// at addresses that are only element-aligned (legal for global accesses) and the
// row's last vector holds n < V valid elements -- those are moved one by one, the
// rest reads as zero and is not written.
template <bool RAG, typename T, int V>
__device__ __forceinline__ void vl(int n, const T *p, T (&v)[V]) {
  if constexpr (!RAG) {
    vload<T, V>(p, v);
  } else {
    if (n >= V) {
      typedef T P __attribute__((ext_vector_type(V), aligned(sizeof(T))));
      const P t = *reinterpret_cast<const P *>(p);
#pragma unroll
      for (int k = 0; k < V; ++k) v[k] = t[k];
    } else {
#pragma unroll
      for (int k = 0; k < V; ++k) v[k] = k < n ? p[k] : T(0);
    }
  }
}

// The same load for a kernel's hot path: the row's last, partial vector is fetched
// whole wherever that stays inside the array (everywhere but near its very end) and
// the elements past the row are zeroed in registers -- no branch, so a wave that holds
// a row end no longer runs the vector and the element-by-element path one after the
// other for every operand.
template <bool RAG, typename T, int V>
__device__ __forceinline__ void vlc(const Voxel &c, const T *p, T (&v)[V]) {
  if constexpr (!RAG) {
    vload<T, V>(p, v);
  } else {
    if (c.nval >= V || c.wide) {
      typedef T P __attribute__((ext_vector_type(V), aligned(sizeof(T))));
      const P t = *reinterpret_cast<const P *>(p);
#pragma unroll
      for (int k = 0; k < V; ++k) v[k] = k < c.nval ? t[k] : T(0);
    } else {
#pragma unroll
      for (int k = 0; k < V; ++k) v[k] = k < c.nval ? p[k] : T(0);
    }
  }
}

template <bool RAG, typename T, int V>
__device__ __forceinline__ void vs(int n, T *p, const T (&v)[V]) {
  if constexpr (!RAG) {
    vstore<T, V>(p, v);
  } else {
    if (n >= V) {
      typedef T P __attribute__((ext_vector_type(V), aligned(sizeof(T))));
      P t;
#pragma unroll
      for (int k = 0; k < V; ++k) t[k] = v[k];
      *reinterpret_cast<P *>(p) = t;
    } else {
#pragma unroll
      for (int k = 0; k < V; ++k)
        if (k < n) p[k] = v[k];
    }
  }
}

// Row groups (ROWS consecutive rows of one z-plane) are dealt to blockIdx.y
// grid-stride: rg = blockIdx.y, blockIdx.y + gridDim.y, ...  A row longer than
// gridDim.x tiles (a 1-D signal of millions of samples) adds an outer x index to
// the units dealt out: unit = row group * x_outer + xo, tile = xo * gridDim.x +
// blockIdx.x.
template <typename T, int VEC, int ROWS>
__device__ __forceinline__ unsigned x_outer(const Geom<T> &G) {
  constexpr int XT = kBlock / ROWS;
  const int64_t tiles = (G.nx + (int64_t)XT * VEC - 1) / ((int64_t)XT * VEC);
  return (unsigned)((tiles + gridDim.x - 1) / gridDim.x);
}

template <typename T, int VEC, int ROWS>
__device__ __forceinline__ int64_t row_groups(const Geom<T> &G) {
  return ((G.ny + ROWS - 1) / ROWS) * G.nz * (int64_t)x_outer<T, VEC, ROWS>(G);
}

template <typename T, int VEC, int ROWS>
__device__ __forceinline__ Voxel voxel_at(const Geom<T> &G, int64_t rg) {
  Voxel c;
  constexpr int XT = kBlock / ROWS;  // threads along x
  const unsigned rgy = (unsigned)((G.ny + ROWS - 1) / ROWS);
  unsigned urg = (unsigned)rg;  // < 2^31 (checked on the host)
  unsigned xo = 0;
  const unsigned nxo = x_outer<T, VEC, ROWS>(G);
  if (nxo > 1) {
    xo = urg % nxo;
    urg /= nxo;
  }
  // Which XCD runs a workgroup is its linear id (blockIdx.x + gridDim.x * blockIdx.y)
  // mod 8, so with row groups dealt in their natural order the groups above and below
  // a workgroup's rows run on OTHER XCDs and every y halo row is a second trip to
  // memory (k_tk1_reg's Lanczos form fetched 1.4x its 12 B per voxel).  Instead the
  // P values of blockIdx.y one XCD cycle spans stand for P slabs of a plane's rows: a
  // slab's groups are then neighbours in one L2, as the groups of the plane above and
  // below already were (same blockIdx.y mod P, a plane of a slab apart).
  unsigned iz = urg / rgy, yg = urg - iz * rgy;
  if (G.slabs && nxo == 1) {
    const unsigned low = gridDim.x & (0u - gridDim.x);       // gcd(gridDim.x, 8) if < 8
    const unsigned sh = low >= 8 ? 0u : (low == 4 ? 1u : (low == 2 ? 2u : 3u));
    const unsigned P = 1u << sh;
    // (every trip of the grid-stride loop must keep blockIdx.y mod P)
    if (sh && rgy % P == 0 &&
        (gridDim.y % P == 0 || (int64_t)gridDim.y >= (int64_t)rgy * G.nz)) {
      const unsigned per = rgy >> sh, r = urg >> sh;
      iz = r / per;
      yg = (urg & (P - 1)) * per + (r - iz * per);
    }
  }
  c.iz = iz;
  c.iy = (int64_t)yg * ROWS + (threadIdx.x / XT);
  c.ix = (((int64_t)xo * gridDim.x + blockIdx.x) * XT + (threadIdx.x % XT)) * VEC;
  c.ok = c.ix < G.nx && c.iy < G.ny;
  c.i = (c.iz * G.ny + c.iy) * G.nx + c.ix;
  const int64_t left = G.nx - c.ix;
  c.nval = left >= VEC ? VEC : (int)left;
  // forward neighbours are read at + sy (2-D, 3-D) and + sz (3-D)
  const int64_t fwd = G.ndim >= 3 ? G.sz + G.sy : (G.ndim == 2 ? G.sy : 0);
  c.wide = c.i + fwd + VEC <= G.n;
  return c;
}

// Workgroups of a stencil kernel's grid: g_stencil_blocks (knob "stencil_blocks", at
// most the kReducePartials partial sums the reduction workspace holds).  The more, the
// better for the kernels that write as much as they read (512^3, 2 048 ... 65 536
// workgroups: k_grad 0.48 / 0.47 / 0.46 / 0.46 / 0.43 / 0.41 ms, the Lanczos update
// 0.60 / 0.51 / 0.46 / 0.45 / 0.45 / 0.45, k_admm_vw 1.18 / 1.08 / 1.07 / 1.03 / 1.00 /
// 0.96; tools/_probe/stencil_time.py); a kernel that only reduces wants fewer (`cap`).
constexpr int kStencilMaxTilesX = 4096;   // x tiles per grid row; longer rows loop

template <int VEC, int ROWS>
inline dim3 stencil_grid(int64_t nz, int64_t ny, int64_t nx, int cap = 0) {
  constexpr int XT = kBlock / ROWS;
  int64_t gx = (nx + (int64_t)XT * VEC - 1) / ((int64_t)XT * VEC);
  if (gx > kStencilMaxTilesX) gx = kStencilMaxTilesX;
  const int64_t nrg = ((ny + ROWS - 1) / ROWS) * nz;
  // (rows that are not whole vectors -- the kernels' ragged forms -- do not gain from
  // more than 16 384: config 4 at 511^3 0.160 - 0.165 s against 0.165 - 0.166 s)
  if (cap <= 0 && VEC > 1 && nx % VEC != 0) cap = 16384;
  const int blocks = cap > 0 && cap < g_stencil_blocks ? cap : g_stencil_blocks;
  int64_t gy = blocks / (gx > 0 ? gx : 1);
  if (gy < 1) gy = 1;
  if (gy > nrg) gy = nrg;
  return dim3((unsigned)gx, (unsigned)gy, 1);
}

// units dealt to blockIdx.y (row groups x outer x index) must fit 31 bits
inline bool stencil_grid_ok(int64_t nz, int64_t ny, int64_t nx) {
  const int64_t outer = (nx / 64 + kStencilMaxTilesX) / kStencilMaxTilesX;
  return ((ny + 3) / 4) * nz * outer < (int64_t)0x7fffffff;
}

template <typename T>
inline bool ptr16(const T *p) {
  return (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}

// Calls f(integral_constant<VEC>, integral_constant<ROWS>) with the widest
// legal vector width; all listed pointers must be 16-byte aligned for VEC > 1.
template <typename T, typename F>
inline int dispatch_stencil(int64_t nz, int64_t ny, int64_t nx, bool aligned,
                            F f) {
  if (!stencil_grid_ok(nz, ny, nx)) return NSOL_EINVAL;
  constexpr int VW = 16 / sizeof(T);
  typedef std::integral_constant<bool, false> No;
  typedef std::integral_constant<bool, true> Yes;
  // `aligned`: every array starts 16-byte aligned.  Otherwise, and for rows that
  // are not a multiple of the vector width, the RAG instantiation runs
  // (element-aligned vectors, ragged tail)
  const bool vec = aligned && (nx % VW == 0);
  const bool rag = !vec && VW > 1 && nx >= 4 * VW;
  if (ny == 1) {
    if (vec) return f(std::integral_constant<int, VW>(), std::integral_constant<int, 1>(), No());
    if (rag) return f(std::integral_constant<int, VW>(), std::integral_constant<int, 1>(), Yes());
    return f(std::integral_constant<int, 1>(), std::integral_constant<int, 1>(), No());
  }
  if (vec) return f(std::integral_constant<int, VW>(), std::integral_constant<int, 4>(), No());
  if (rag) return f(std::integral_constant<int, VW>(), std::integral_constant<int, 4>(), Yes());
  return f(std::integral_constant<int, 1>(), std::integral_constant<int, 4>(), No());
}

// forward difference of the lane's VEC voxels along x/y/z, reference order
// hi*w + lo*(-w); `right` = value just past the vector (0 at the boundary)
template <typename T, int VEC>
__device__ __forceinline__ void fwd_diff_x(const T (&c)[VEC], T right, T w,
                                           T (&g)[VEC]) {
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    const T hi = (k + 1 < VEC) ? c[(k + 1) % VEC] : right;
    g[k] = hi * w + c[k] * (-w);
  }
}

template <typename T, int VEC>
__device__ __forceinline__ void fwd_diff(const T (&c)[VEC], const T (&hi)[VEC],
                                         T w, T (&g)[VEC]) {
#pragma unroll
  for (int k = 0; k < VEC; ++k) g[k] = hi[k] * w + c[k] * (-w);
}

}  // namespace nsol
