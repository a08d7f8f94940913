// Persistent primal-dual kernel for cache-resident volumes (BASELINE configs 1-2:
// 256^2 x 50 iterations, 64^3 x 200 iterations): ONE launch runs the whole
// Chambolle-Pock loop of primal_dual_solver.py:232-261.
//
// With one launch per iteration such a volume costs 7-8 us per iteration -- the
// latency of a short kernel plus a launch boundary -- for about a microsecond of
// arithmetic and traffic.  A grid barrier per iteration would cost the same again
// (MI355X_MICROARCH.md prices the XCD-hierarchical barrier at 4.1 us for 256
// workgroups).  Here every workgroup owns a tile of the volume for the whole run
// and keeps its x, xbar, b~ and p in REGISTERS (one 16-byte vector per lane and
// array); per iteration it only
//   * exchanges its own values between lanes through LDS (two barriers), and
//   * hands the one-voxel faces of xbar (first and last layer per axis) and of p
//     (last layer, the component of that axis) to its <= 6 face neighbours through
//     global memory: the faces are written with agent-scope (sc1, write-through)
//     stores, drained (s_waitcnt vmcnt(0) + workgroup barrier), then ONE flag per
//     tile is raised to the iteration number; a neighbour polls that flag with one
//     lane, and then reads the faces with sc1 loads.  Two slots per face by
//     iteration parity -- a workgroup can only run one iteration ahead of a
//     neighbour, because it needs that neighbour's faces of every iteration.
//     (A first version tagged every value, {value, tag} in 8 bytes, and polled the
//     values themselves: one round trip less per iteration, but 27 tiny requests per
//     boundary lane and round -- 14 us per iteration at 64^3, against 8 with one
//     launch per iteration.)
// The arithmetic per voxel is that of k_pd_fused in the same order (the new dual of
// the lower neighbour voxel across a tile face is recomputed from the neighbour's
// old p and xbar, exactly as k_pd_fused recomputes it across its patches), so the
// result is bit-identical to one launch of k_pd_fused per iteration.
//
// Co-residency: the grid is at most one workgroup per CU (checked on the host, against
// the occupancy the runtime reports for this kernel and launch shape), so every
// workgroup is running when its neighbours wait for it.  On a shared or CU-masked
// device that can still fail, so every wait is bounded: a workgroup that does not see
// a flag within ~2^21 polls gives up polling for the rest of the run and raises the
// error word, which the host reads after the run -- no wave can spin forever.  The
// kernel READS the state from (xbar, x, p) and WRITES the result to (xbar_out,
// x_out, p_out): with distinct output arrays the inputs survive a failed run and the
// caller repeats it with one launch per iteration (nsol_amd/ops.py does).
#include <string.h>

#include "nsol_common.hpp"
#include "nsol_pd_common.hpp"

using namespace nsol;

namespace {

constexpr int kMaxTiles = 256;
constexpr int kMaxSpin = 1 << 21;   // polling rounds of ~1-2 us before a workgroup gives up
constexpr int kKinds = 3;          // XF: xbar first layer, XL: xbar last, PL: p last
// debug knobs (nsol_hip_set_param_pdp): "pdp_max_spin" bounds the polls of a wait,
// "pdp_mute_tile" names a tile that never raises its flag -- together they force the
// time-out path in a test without oversubscribing the device
int g_max_spin = kMaxSpin;
int g_mute_tile = -1;

template <typename T>
struct IterScalars {               // one Chambolle-Pock iteration's step sizes
  T sigma, hden, tau, tl, one_plus_tl, theta;
};

struct Tiling {
  int lx, ly, lz;                  // lanes per tile along x (vectors), y, z
  int ntx, nty, ntz;
  int face;                        // elements of the largest tile face
};

// agent-scope relaxed accesses (sc1): written through to / read from the level all
// XCDs share, in 8-byte pieces where the data allow
__device__ __forceinline__ void put1(float *g, float v) {
  __hip_atomic_store(g, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void put1(double *g, double v) {
  __hip_atomic_store(g, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float get1(const float *g) {
  return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double get1(const double *g) {
  return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void putv(float *g, const float (&v)[4]) {
  uint64_t *q = reinterpret_cast<uint64_t *>(g);
#pragma unroll
  for (int h = 0; h < 2; ++h)
    __hip_atomic_store(q + h, ((uint64_t)__float_as_uint(v[2 * h + 1]) << 32) |
                                  (uint64_t)__float_as_uint(v[2 * h]),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void putv(double *g, const double (&v)[2]) {
  put1(g, v[0]);
  put1(g + 1, v[1]);
}
__device__ __forceinline__ void getv(const float *g, float (&v)[4]) {
  const uint64_t *q = reinterpret_cast<const uint64_t *>(g);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint64_t w = __hip_atomic_load(q + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v[2 * h] = __uint_as_float((uint32_t)w);
    v[2 * h + 1] = __uint_as_float((uint32_t)(w >> 32));
  }
}
__device__ __forceinline__ void getv(const double *g, double (&v)[2]) {
  v[0] = get1(g);
  v[1] = get1(g + 1);
}

// MAXT: upper bound of the lanes per tile (the 256-lane form has registers for a
// boundary lane's 27-30 granules in flight; the 1024-lane form spills a few of them)
template <typename T, int VEC, int NDIM, int MAXT>
__global__ __launch_bounds__(MAXT) void k_pd_persist(
    const T *xbar, const T *x, const T *__restrict__ bt, const T *p, T *xbar_out, T *x_out,
    T *p_out, Geom<T> G, const IterScalars<T> *__restrict__ sc, int iterations, int huber,
    int l1, int has_p, Tiling Q, T *__restrict__ halo, unsigned int *__restrict__ flags,
    unsigned int *__restrict__ err, int max_spin, int mute_tile) {
  typedef typename Pack<T, VEC>::type V;
  extern __shared__ __attribute__((aligned(16))) unsigned char pdp_smem[];
  const int nthr = Q.lx * Q.ly * Q.lz;
  V *sx = reinterpret_cast<V *>(pdp_smem);          // xbar of the tile
  V *spy = sx + nthr;                               // new p_y
  V *spz = spy + nthr;                              // new p_z
  T *spx = reinterpret_cast<T *>(spz + nthr);       // last element of new p_x
  const int tid = threadIdx.x;
  const int lx = tid % Q.lx, ly = (tid / Q.lx) % Q.ly, lz = tid / (Q.lx * Q.ly);
  int bid = blockIdx.x;
  const int tx = bid % Q.ntx; bid /= Q.ntx;
  const int ty = bid % Q.nty;
  const int tz = bid / Q.nty;
  const int64_t x0 = ((int64_t)tx * Q.lx + lx) * VEC;
  const int64_t y = (int64_t)ty * Q.ly + ly, z = (int64_t)tz * Q.lz + lz;
  const bool in = x0 < G.nx && y < G.ny && z < G.nz;
  const int64_t off = z * G.sz + y * G.sy + x0;
  // neighbour voxels inside the volume / across a tile face
  const bool has_r = in && x0 + VEC < G.nx, has_l = in && x0 > 0;
  const bool has_u = in && NDIM >= 2 && y + 1 < G.ny, has_d = in && NDIM >= 2 && y > 0;
  const bool has_b = in && NDIM >= 3 && z + 1 < G.nz, has_f = in && NDIM >= 3 && z > 0;
  const bool edge_r = lx == Q.lx - 1, edge_l = lx == 0;
  const bool edge_u = ly == Q.ly - 1, edge_d = ly == 0;
  const bool edge_b = lz == Q.lz - 1, edge_f = lz == 0;
  // faces: [tile][slot][axis][kind][face element]; flags: [tile]
  const int64_t face = Q.face;
  constexpr int XF = 0, XL = 1, PL = 2;              // kinds
  auto tile_id = [&](int ttx, int tty, int ttz) {
    return ((int64_t)ttz * Q.nty + tty) * Q.ntx + ttx;
  };
  auto hp = [&](int64_t t, int slot, int axis, int kind, int64_t idx) {
    return halo + (((t * 2 + slot) * 3 + axis) * kKinds + kind) * face + idx;
  };
  const int64_t me = tile_id(tx, ty, tz);
  // face element index of this lane's voxels on the faces normal to each axis
  const int64_t fx = (int64_t)lz * Q.ly + ly;                        // one value
  const int64_t fy = ((int64_t)lz * Q.lx + lx) * VEC;                // VEC values
  const int64_t fz = ((int64_t)ly * Q.lx + lx) * VEC;                // VEC values

  T xb[VEC], xv[VEC], bv[VEC], px[VEC], py[VEC], pz[VEC];
  zero(xb); zero(xv); zero(bv); zero(px); zero(py); zero(pz);
  if (in) {
    ldv<T, VEC>(xbar + off, xb);
    ldv<T, VEC>(x + off, xv);
    ldv<T, VEC>(bt + off, bv);
    if (has_p) {
      ldv<T, VEC>(p + off, px);
      if constexpr (NDIM >= 2) ldv<T, VEC>(p + G.n + off, py);
      if constexpr (NDIM >= 3) ldv<T, VEC>(p + 2 * G.n + off, pz);
    }
  }
  bool failed = false;
  // faces of the state after `k` iterations go into slot k & 1; the tile's flag is
  // raised to k + 1 once every lane's stores have been acknowledged
  auto publish = [&](int k) {
    const int s = k & 1;
    if (in) {
      if (edge_l && has_l) put1(hp(me, s, 0, XF, fx), xb[0]);
      if (edge_r && has_r) {
        put1(hp(me, s, 0, XL, fx), xb[VEC - 1]);
        put1(hp(me, s, 0, PL, fx), px[VEC - 1]);
      }
      if constexpr (NDIM >= 2) {
        if (edge_d && has_d) putv(hp(me, s, 1, XF, fy), xb);
        if (edge_u && has_u) {
          putv(hp(me, s, 1, XL, fy), xb);
          putv(hp(me, s, 1, PL, fy), py);
        }
      }
      if constexpr (NDIM >= 3) {
        if (edge_f && has_f) putv(hp(me, s, 2, XF, fz), xb);
        if (edge_b && has_b) {
          putv(hp(me, s, 2, XL, fz), xb);
          putv(hp(me, s, 2, PL, fz), pz);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && me != (int64_t)mute_tile)
      __hip_atomic_store(flags + me, (unsigned int)k + 1u, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
  };
  publish(0);
  // the (up to six) neighbour tiles whose flags lanes 0..5 poll
  int64_t nb_tile = -1;
  if (tid < 6) {
    const int axis = tid >> 1, up = tid & 1;
    const int ntile[3] = {Q.ntx, Q.nty, Q.ntz};
    const int t3[3] = {tx, ty, tz};
    if (axis < NDIM) {
      int c[3] = {tx, ty, tz};
      c[axis] += up ? 1 : -1;
      // a neighbour tile matters only if it holds voxels of the volume
      const int64_t lanes[3] = {(int64_t)Q.lx * VEC, Q.ly, Q.lz};
      const int64_t ext[3] = {G.nx, G.ny, G.nz};
      if (c[axis] >= 0 && c[axis] < ntile[axis] && (int64_t)c[axis] * lanes[axis] < ext[axis])
        nb_tile = tile_id(c[0], c[1], c[2]);
      (void)t3;
    }
  }

  for (int k = 0; k < iterations; ++k) {
    PdScalars<T> S;
    {
      const IterScalars<T> c = sc[k];
      S.sigma = c.sigma; S.hden = c.hden; S.tau = c.tau; S.tl = c.tl;
      S.one_plus_tl = c.one_plus_tl; S.theta = c.theta;
      S.huber = huber; S.l1 = l1; S.has_p = 1;
    }
    const int s = k & 1;
    // ---- wait until every neighbour tile has published its state after k
    //      iterations (flag >= k + 1), then read its faces
    if (nb_tile >= 0 && !failed) {
      int spin = 0;
      while (__hip_atomic_load(flags + nb_tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <
             (unsigned int)k + 1u) {
        if (++spin >= max_spin) { failed = true; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    T xr = T(0), xl = T(0), pl_old = T(0);
    T xu[VEC], xd[VEC], pd_old[VEC], xbk[VEC], xf[VEC], pf_old[VEC];
    zero(xu); zero(xd); zero(pd_old); zero(xbk); zero(xf); zero(pf_old);
    if (edge_r && has_r) xr = get1(hp(tile_id(tx + 1, ty, tz), s, 0, XF, fx));
    if (edge_l && has_l) {
      const int64_t t = tile_id(tx - 1, ty, tz);
      xl = get1(hp(t, s, 0, XL, fx));
      pl_old = get1(hp(t, s, 0, PL, fx));
    }
    if constexpr (NDIM >= 2) {
      if (edge_u && has_u) getv(hp(tile_id(tx, ty + 1, tz), s, 1, XF, fy), xu);
      if (edge_d && has_d) {
        const int64_t t = tile_id(tx, ty - 1, tz);
        getv(hp(t, s, 1, XL, fy), xd);
        getv(hp(t, s, 1, PL, fy), pd_old);
      }
    }
    if constexpr (NDIM >= 3) {
      if (edge_b && has_b) getv(hp(tile_id(tx, ty, tz + 1), s, 2, XF, fz), xbk);
      if (edge_f && has_f) {
        const int64_t t = tile_id(tx, ty, tz - 1);
        getv(hp(t, s, 2, XL, fz), xf);
        getv(hp(t, s, 2, PL, fz), pf_old);
      }
    }
    // ---- xbar of the tile to LDS; upper neighbours of every voxel
    {
      V t;
#pragma unroll
      for (int e = 0; e < VEC; ++e) t[e] = xb[e];
      sx[tid] = t;
    }
    __syncthreads();
    T nbx = T(0);                                   // xbar right of the vector
    if (has_r) nbx = edge_r ? xr : sx[tid + 1][0];
    T hy[VEC], hz[VEC];
    zero(hy); zero(hz);
    if constexpr (NDIM >= 2) {
      if (has_u) {
        if (edge_u) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) hy[e] = xu[e];
        } else {
          const V t = sx[tid + Q.lx];
#pragma unroll
          for (int e = 0; e < VEC; ++e) hy[e] = t[e];
        }
      }
    }
    if constexpr (NDIM >= 3) {
      if (has_b) {
        if (edge_b) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) hz[e] = xbk[e];
        } else {
          const V t = sx[tid + Q.lx * Q.ly];
#pragma unroll
          for (int e = 0; e < VEC; ++e) hz[e] = t[e];
        }
      }
    }
    // ---- dual update at the lane's own voxels (k_pd_fused's order)
    T pxn[VEC], pyn[VEC], pzn[VEC];
    zero(pyn); zero(pzn);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const T hi = (e + 1 < VEC) ? xb[(e + 1) % VEC] : nbx;
      pxn[e] = dual_update(px[e], hi, xb[e], G.wx, S);
    }
    if constexpr (NDIM >= 2) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) pyn[e] = dual_update(py[e], hy[e], xb[e], G.wy, S);
    }
    if constexpr (NDIM >= 3) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) pzn[e] = dual_update(pz[e], hz[e], xb[e], G.wz, S);
    }
    // ---- new dual values of the lower neighbours: inside the tile through LDS,
    //      across a face recomputed from the neighbour's old p and xbar
    spx[tid] = pxn[VEC - 1];
    if constexpr (NDIM >= 2) {
      V t;
#pragma unroll
      for (int e = 0; e < VEC; ++e) t[e] = pyn[e];
      spy[tid] = t;
    }
    if constexpr (NDIM >= 3) {
      V t;
#pragma unroll
      for (int e = 0; e < VEC; ++e) t[e] = pzn[e];
      spz[tid] = t;
    }
    __syncthreads();
    T pxl = T(0);
    if (has_l) pxl = edge_l ? dual_update(pl_old, xb[0], xl, G.wx, S) : spx[tid - 1];
    T pyu[VEC], pzp[VEC];
    zero(pyu); zero(pzp);
    if constexpr (NDIM >= 2) {
      if (has_d) {
        if (edge_d) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) pyu[e] = dual_update(pd_old[e], xb[e], xd[e], G.wy, S);
        } else {
          const V t = spy[tid - Q.lx];
#pragma unroll
          for (int e = 0; e < VEC; ++e) pyu[e] = t[e];
        }
      }
    }
    if constexpr (NDIM >= 3) {
      if (has_f) {
        if (edge_f) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) pzp[e] = dual_update(pf_old[e], xb[e], xf[e], G.wz, S);
        } else {
          const V t = spz[tid - Q.lx * Q.ly];
#pragma unroll
          for (int e = 0; e < VEC; ++e) pzp[e] = t[e];
        }
      }
    }
    // ---- primal update
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const T pl = (e > 0) ? pxn[(e + VEC - 1) % VEC] : pxl;
      T kt = pxn[e] * (-G.wx) + pl * G.wx;
      if constexpr (NDIM >= 2) kt += pyn[e] * (-G.wy) + pyu[e] * G.wy;
      if constexpr (NDIM >= 3) kt += pzn[e] * (-G.wz) + pzp[e] * G.wz;
      const T u = xv[e] - S.tau * kt;
      const T xnew = prox_data(u, bv[e], S.tl, S.one_plus_tl, S.l1 != 0);
      xb[e] = xnew + S.theta * (xnew - xv[e]);
      xv[e] = xnew;
      px[e] = pxn[e];
      py[e] = pyn[e];
      pz[e] = pzn[e];
    }
    if (k + 1 < iterations) publish(k + 1);
    // (the LDS tiles are rewritten after the next iteration's first barrier only
    // by lanes that have passed this iteration's second one: no third barrier)
  }
  if (in) {
    stv<T, VEC>(xbar_out + off, xb);
    stv<T, VEC>(x_out + off, xv);
    stv<T, VEC>(p_out + off, px);
    if constexpr (NDIM >= 2) stv<T, VEC>(p_out + G.n + off, py);
    if constexpr (NDIM >= 3) stv<T, VEC>(p_out + 2 * G.n + off, pz);
  }
  if (failed) atomicOr(err, 1u);
}

// Step sizes of up to kSetupChunk iterations travel as kernel arguments (no host
// -> device copy to wait for) and are turned into the per-iteration scalars on the
// device with the arithmetic of make_scalars() (IEEE double, no contraction: the
// same bits as on the host); the first chunk also clears the tiles' flags.
constexpr int kSetupChunk = 128;
struct SetupArgs {
  double sig[kSetupChunk], tau[kSetupChunk], theta[kSetupChunk];
};
__device__ __forceinline__ float dev_huber_den(double den, float) { return (float)(1.0 / den); }
__device__ __forceinline__ double dev_huber_den(double den, double) { return den; }
__device__ __forceinline__ float dev_prox_den(double tl, float) { return (float)(1.0 / (1.0 + tl)); }
__device__ __forceinline__ double dev_prox_den(double tl, double) { return 1.0 + tl; }

template <typename T>
__global__ __launch_bounds__(kBlock) void k_pdp_setup(SetupArgs A, int first, int count,
                                                       double lambda, double gamma,
                                                       int huber, IterScalars<T> *sc,
                                                       unsigned int *flags, int nflags) {
  const int i = threadIdx.x;
  if (i < count) {
    const double hden = huber ? 1.0 + A.sig[i] * gamma : 1.0;
    const double tl = A.tau[i] * lambda;
    IterScalars<T> c;
    c.sigma = (T)A.sig[i]; c.hden = dev_huber_den(hden, T(0)); c.tau = (T)A.tau[i];
    c.tl = (T)tl; c.one_plus_tl = dev_prox_den(tl, T(0)); c.theta = (T)A.theta[i];
    sc[first + i] = c;
  }
  if (first == 0)
    for (int j = i; j < nflags; j += kBlock) flags[j] = 0u;
}

inline int cu_count_pdp() {
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

// lanes per tile: powers of two with 64..1024 lanes that cover the volume in at
// most one tile per CU; among those the most tiles (the shortest latency chain per
// iteration), then the fewest lanes
template <int VEC>
bool pick_tiling(int ndim, int64_t nz, int64_t ny, int64_t nx, Tiling *out) {
  const int64_t nxv = nx / VEC;
  const int max_tiles = cu_count_pdp() < kMaxTiles ? cu_count_pdp() : kMaxTiles;
  int64_t best_tiles = -1, best_lanes = 0;
  for (int lx = 1; lx <= 256; lx *= 2)
    for (int ly = 1; ly <= (ndim >= 2 ? 256 : 1); ly *= 2)
      for (int lz = 1; lz <= (ndim >= 3 ? 256 : 1); lz *= 2) {
        const int64_t lanes = (int64_t)lx * ly * lz;
        if (lanes < 64 || lanes > 1024) continue;
        if ((lx > 1 && lx / 2 >= nxv) || (ly > 1 && ly / 2 >= ny) || (lz > 1 && lz / 2 >= nz))
          continue;                                   // (a smaller box covers the axis)
        const int64_t ntx = (nxv + lx - 1) / lx, nty = (ny + ly - 1) / ly,
                      ntz = (nz + lz - 1) / lz;
        const int64_t tiles = ntx * nty * ntz;
        if (tiles > max_tiles) continue;
        if (tiles > best_tiles || (tiles == best_tiles && lanes < best_lanes)) {
          best_tiles = tiles; best_lanes = lanes;
          out->lx = lx; out->ly = ly; out->lz = lz;
          out->ntx = (int)ntx; out->nty = (int)nty; out->ntz = (int)ntz;
        }
      }
  if (best_tiles < 0) return false;
  const int fx = out->ly * out->lz, fy = out->lx * VEC * out->lz, fz = out->lx * VEC * out->ly;
  out->face = fx > fy ? (fx > fz ? fx : fz) : (fy > fz ? fy : fz);
  return true;
}

template <typename T>
int64_t persist_ws_bytes(int ndim, int64_t nz, int64_t ny, int64_t nx, int iterations) {
  constexpr int VEC = 16 / (int)sizeof(T);
  Tiling Q;
  if (!geom_ok(ndim, nz, ny, nx) || nx % VEC != 0 || iterations < 1 ||
      !pick_tiling<VEC>(ndim, nz, ny, nx, &Q))
    return -1;
  const int64_t tiles = (int64_t)Q.ntx * Q.nty * Q.ntz;
  const int64_t halo = tiles * 2 * 3 * kKinds * Q.face * (int64_t)sizeof(T);
  const int64_t scal = (((int64_t)iterations * sizeof(IterScalars<T>)) + 255) / 256 * 256;
  const int64_t flg = (tiles * 4 + 255) / 256 * 256;
  return 256 + scal + flg + halo;                  // [error word | scalars | flags | faces]
}

template <typename T>
int persist_run(const T *xbar, const T *x, const T *bt, const T *p, T *xbar_out, T *x_out,
                T *p_out, int ndim, int64_t nz, int64_t ny,
                int64_t nx, double wx, double wy, double wz, double lambda,
                const double *sig, const double *tau, const double *theta, int iterations,
                int p_is_zero, double gamma_huber, int flags, void *ws, int64_t ws_bytes,
                unsigned int *err_word, void *stream) {
  NSOL_CHECK_GEOM(ndim, nz, ny, nx);
  constexpr int VEC = 16 / (int)sizeof(T);
  if (!xbar || !x || !bt || !p || !xbar_out || !x_out || !p_out || !sig || !tau ||
      !theta || iterations < 1 || !ws)
    return NSOL_EINVAL;
  const int64_t need = persist_ws_bytes<T>(ndim, nz, ny, nx, iterations);
  if (need < 0) return -2;                          // the kernel does not apply
  if (ws_bytes < need || ((uintptr_t)ws & 15u)) return NSOL_EINVAL;
  if (((uintptr_t)xbar | (uintptr_t)x | (uintptr_t)bt | (uintptr_t)p | (uintptr_t)xbar_out |
       (uintptr_t)x_out | (uintptr_t)p_out) & 15u)
    return -2;
  if ((nz * ny * nx) % VEC != 0) return -2;
  Tiling Q;
  pick_tiling<VEC>(ndim, nz, ny, nx, &Q);
  const Geom<T> G = make_geom<T>(ndim, nz, ny, nx, wx, wy, wz);
  const bool huber = (flags & NSOL_PD_REG_HUBER) != 0;
  hipStream_t st = as_stream(stream);
  unsigned char *base = static_cast<unsigned char *>(ws);
  const int64_t scal = (((int64_t)iterations * sizeof(IterScalars<T>)) + 255) / 256 * 256;
  const int64_t tiles = (int64_t)Q.ntx * Q.nty * Q.ntz;
  const int64_t flg = (tiles * 4 + 255) / 256 * 256;
  auto *scd = reinterpret_cast<IterScalars<T> *>(base + 256);
  auto *flags_d = reinterpret_cast<unsigned int *>(base + 256 + scal);
  auto *halo = reinterpret_cast<T *>(base + 256 + scal + flg);
  // the error word: the caller's (e.g. pinned host memory it can look at after its
  // own synchronisation) or the first word of the workspace
  unsigned int *err = err_word ? err_word : reinterpret_cast<unsigned int *>(base);
  if (!err_word) {
    hipError_t e = hipMemsetAsync(base, 0, 256, st);
    if (e != hipSuccess) return (int)e;
  }
  for (int first = 0; first < iterations; first += kSetupChunk) {
    SetupArgs A;
    const int count = iterations - first < kSetupChunk ? iterations - first : kSetupChunk;
    for (int i = 0; i < count; ++i) {
      A.sig[i] = sig[first + i]; A.tau[i] = tau[first + i]; A.theta[i] = theta[first + i];
    }
    for (int i = count; i < kSetupChunk; ++i) A.sig[i] = A.tau[i] = A.theta[i] = 0.0;
    hipLaunchKernelGGL(k_pdp_setup<T>, dim3(1), dim3(kBlock), 0, st, A, first, count, lambda,
                       gamma_huber, huber ? 1 : 0, scd, flags_d, (int)tiles);
  }
  const int nthr = Q.lx * Q.ly * Q.lz;
  const size_t lds = (size_t)nthr * (3 * 16 + sizeof(T));
  const dim3 grid((unsigned)(Q.ntx * Q.nty * Q.ntz));
#define NSOL_PDP_GO(ND, MT)                                                          \
  do {                                                                               \
    /* one workgroup per tile must be resident at once: ask the runtime what this   \
       kernel's registers and LDS admit per CU (advisory -- hence the bounded waits) */ \
    int per_cu = 0;                                                                  \
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(                               \
            &per_cu, k_pd_persist<T, VEC, ND, MT>, nthr, lds) != hipSuccess ||       \
        (int64_t)per_cu * cu_count_pdp() < (int64_t)grid.x)                          \
      return -2;                                                                     \
    hipLaunchKernelGGL((k_pd_persist<T, VEC, ND, MT>), grid, dim3(nthr), lds, st, xbar, x, \
                       bt, p, xbar_out, x_out, p_out, G, scd, iterations, huber ? 1 : 0, \
                       (flags & NSOL_PD_DATA_L1) ? 1 : 0, p_is_zero ? 0 : 1, Q, halo,    \
                       flags_d, err, g_max_spin, g_mute_tile);                       \
  } while (0)
  if (nthr <= 256) {
    switch (ndim) {
      case 1: NSOL_PDP_GO(1, 256); break;
      case 2: NSOL_PDP_GO(2, 256); break;
      default: NSOL_PDP_GO(3, 256); break;
    }
  } else {
    switch (ndim) {
      case 1: NSOL_PDP_GO(1, 1024); break;
      case 2: NSOL_PDP_GO(2, 1024); break;
      default: NSOL_PDP_GO(3, 1024); break;
    }
  }
#undef NSOL_PDP_GO
  return launch_status();
}

}  // namespace

extern "C" {

int64_t nsol_pd_persist_ws_bytes(int elem_size, int ndim, int64_t nz, int64_t ny,
                                 int64_t nx, int iterations) {
  if (elem_size == 4) return persist_ws_bytes<float>(ndim, nz, ny, nx, iterations);
  if (elem_size == 8) return persist_ws_bytes<double>(ndim, nz, ny, nx, iterations);
  return -1;
}

#define NSOL_PDP_DEF(T, SUF)                                                        \
  int nsol_pd_persist_run_to_##SUF(                                                 \
      const T *xbar, const T *x, const T *bt, const T *p, T *xbar_out, T *x_out,   \
      T *p_out, int ndim, int64_t nz, int64_t ny, int64_t nx, double wx, double wy, \
      double wz, double lambda, const double *sigma_host, const double *tau_host,  \
      const double *theta_host, int iterations, int p_is_zero, double gamma_huber, \
      int flags, void *ws, int64_t ws_bytes, unsigned int *err_word, void *stream) { \
    return persist_run<T>(xbar, x, bt, p, xbar_out, x_out, p_out, ndim, nz, ny, nx, \
                          wx, wy, wz, lambda, sigma_host, tau_host, theta_host,     \
                          iterations, p_is_zero, gamma_huber, flags, ws, ws_bytes,  \
                          err_word, stream);                                        \
  }                                                                                 \
  int nsol_pd_persist_run_##SUF(                                                    \
      T *xbar, T *x, const T *bt, T *p, int ndim, int64_t nz, int64_t ny,           \
      int64_t nx, double wx, double wy, double wz, double lambda,                   \
      const double *sigma_host, const double *tau_host, const double *theta_host,  \
      int iterations, int p_is_zero, double gamma_huber, int flags, void *ws,       \
      int64_t ws_bytes, unsigned int *err_word, void *stream) {                     \
    return persist_run<T>(xbar, x, bt, p, xbar, x, p, ndim, nz, ny, nx, wx, wy, wz, \
                          lambda, sigma_host, tau_host, theta_host, iterations,     \
                          p_is_zero, gamma_huber, flags, ws, ws_bytes, err_word,    \
                          stream);                                                  \
  }
NSOL_PDP_DEF(float, f32)
NSOL_PDP_DEF(double, f64)
#undef NSOL_PDP_DEF

/* debug knobs: "pdp_max_spin" (polls before a wait gives up), "pdp_mute_tile" (a
 * tile that never raises its flag; -1 = none) */
int nsol_hip_set_param_pdp(const char *name, int value) {
  if (!name) return NSOL_EINVAL;
  if (!strcmp(name, "pdp_max_spin")) g_max_spin = value < 1 ? 1 : value;
  else if (!strcmp(name, "pdp_mute_tile")) g_mute_tile = value;
  else return NSOL_EINVAL;
  return 0;
}

}  // extern "C"
