// Chambolle-Pock primal-dual iteration for gfx950 (MI355X).
//
// reference: primal_dual_solver.py:232-261 driving linear_operators.py:121-169
// and proximal_operators.py:95-159.
//
// Three forms, all producing the same per-voxel arithmetic (same operation
// order as the NumPy reference; the library is built with -ffp-contract=off):
//   * k_dual_step / k_primal_step: two passes (14 words per voxel in 3-D);
//   * k_pd_fused: ONE pass, 11 words per voxel (read xbar, x, bt, p[3]; write
//     p[3], x, xbar).  Each wave owns a (LX*VEC) x (LY*RY) patch of an x-y tile
//     and marches along z, keeping xbar[z], xbar[z+1] and the new p_z[z-1] in
//     registers; x-neighbours travel by wave shuffles, y-neighbours live in
//     registers (RY rows per lane) or come from the neighbouring lane row; only
//     the patch's outer halo (one row above/below, one column left/right) is
//     re-read from L1/L2.  The dual update of the lower halo (p_new at i - e_a)
//     is recomputed instead of being exchanged, so there is no inter-workgroup
//     dependency inside a launch; xbar and p are ping-pong buffers.
#include <stdlib.h>
#include <string.h>

#include "nsol_common.hpp"
#include "nsol_pd_common.hpp"

using namespace nsol;

namespace {

struct PdTuning {
  int zchunk = 0;   // 0 = auto
  int ry = 0;       // rows per lane (1, 2 or 4); 0 = auto
  int force_two_pass = 0;
  int xcd_map = 1;  // 0 = plain block order, 1 = XCD-aware slabs
  int rag = 1;      // unaligned rows: 1 = element-aligned 16-byte accesses, 0 = 4-byte
};
PdTuning g_tune;

// ---------------------------------------------------------------------------
// two-pass form
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void k_dual_step(
    const T *__restrict__ xbar, const T *p_in, T *p_out, Geom<T> G, T sigma,
    T hden, bool huber) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < G.n;
       i += stride) {
    const int64_t ix = i % G.nx;
    const int64_t r = i / G.nx;
    const int64_t iy = r % G.ny;
    const int64_t iz = r / G.ny;
    const T c = xbar[i];
    {
      const T nb = (ix + 1 < G.nx) ? xbar[i + 1] : T(0);
      T q = (p_in ? p_in[i] : T(0)) + sigma * (nb * G.wx + c * (-G.wx));
      if (huber) q = huber_div(q, hden);
      p_out[i] = dual_clamp(q);
    }
    if (G.ndim >= 2) {
      const T nb = (iy + 1 < G.ny) ? xbar[i + G.sy] : T(0);
      T q = (p_in ? p_in[G.n + i] : T(0)) + sigma * (nb * G.wy + c * (-G.wy));
      if (huber) q = huber_div(q, hden);
      p_out[G.n + i] = dual_clamp(q);
    }
    if (G.ndim >= 3) {
      const T nb = (iz + 1 < G.nz) ? xbar[i + G.sz] : T(0);
      T q = (p_in ? p_in[2 * G.n + i] : T(0)) +
            sigma * (nb * G.wz + c * (-G.wz));
      if (huber) q = huber_div(q, hden);
      p_out[2 * G.n + i] = dual_clamp(q);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_primal_step(
    const T *__restrict__ p, T *__restrict__ x, T *__restrict__ xbar,
    const T *__restrict__ bt, Geom<T> G, T tau, T tl, T one_plus_tl, T theta,
    bool l1) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < G.n;
       i += stride) {
    const int64_t ix = i % G.nx;
    const int64_t r = i / G.nx;
    T kt = p[i] * (-G.wx) + ((ix > 0) ? p[i - 1] : T(0)) * G.wx;
    if (G.ndim >= 2) {
      const T *py = p + G.n;
      kt += py[i] * (-G.wy) + ((r % G.ny > 0) ? py[i - G.sy] : T(0)) * G.wy;
    }
    if (G.ndim >= 3) {
      const T *pz = p + 2 * G.n;
      kt += pz[i] * (-G.wz) + ((r / G.ny > 0) ? pz[i - G.sz] : T(0)) * G.wz;
    }
    const T xo = x[i];
    const T u = xo - tau * kt;
    const T xn = prox_data(u, bt[i], tl, one_plus_tl, l1);
    x[i] = xn;
    xbar[i] = xn + theta * (xn - xo);
  }
}

// ---------------------------------------------------------------------------
// single-pass fused form
// ---------------------------------------------------------------------------
// RAG: rows that are not a multiple of VEC elements / arrays that are not 16-byte
// aligned (ldv_rag / stv_rag; the elements of a row's last vector that lie behind
// the row read as zero -- exactly the "u := 0 past the last index" of the forward
// difference -- and are never stored).
template <typename T, int VEC, int LX, int RY, int NDIM, bool RAG = false>
__global__ __launch_bounds__(kBlock) void k_pd_fused(
    const T *__restrict__ xbar_in, T *__restrict__ xbar_out, T *x,
    const T *__restrict__ bt, const T *__restrict__ p_in,
    T *__restrict__ p_out, Geom<T> G, PdScalars<T> S, int ntx, int nty,
    int zchunk, int slab) {
  constexpr int LY = kWave / LX;
  constexpr int WAVES = kBlock / kWave;
  constexpr int TY = WAVES * LY * RY;

  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int lx = lane % LX;
  const int ly = lane / LX;
  // Block -> tile map.  Workgroups are dealt round-robin to the 8 XCDs
  // (blockIdx % 8 names the group that shares an L2), so with slab > 0 each
  // XCD walks its own slab of `slab` consecutive y-tiles: the halo rows and
  // columns that neighbouring tiles re-read are then served by that XCD's L2.
  // Placement only affects speed, never results.
  int tx, ty, zc;
  if (slab > 0) {
    const int xcd = blockIdx.x & 7;
    int j = blockIdx.x >> 3;
    tx = j % ntx;
    j /= ntx;
    ty = xcd * slab + j % slab;
    zc = j / slab;
    if (ty >= nty) return;
  } else {
    int bid = blockIdx.x;
    tx = bid % ntx;
    bid /= ntx;
    ty = bid % nty;
    zc = bid / nty;
  }

  const int64_t x0 = ((int64_t)tx * LX + lx) * VEC;
  const int64_t y0 = (int64_t)ty * TY + (int64_t)(wave * LY + ly) * RY;
  const bool xin = x0 < G.nx;
  // valid elements of this lane's vector (RAG: the row may end inside it)
  const int nval = !RAG ? VEC : (G.nx - x0 >= VEC ? VEC : (int)(xin ? G.nx - x0 : 0));
  auto ld = [&](const T *q, T (&v)[VEC]) {
    if constexpr (RAG) ldv_rag<T, VEC>(q, v, nval);
    else ldv<T, VEC>(q, v);
  };
  auto st = [&](T *q, const T (&v)[VEC]) {
    if constexpr (RAG) stv_rag<T, VEC>(q, v, nval);
    else stv<T, VEC>(q, v);
  };
  bool rin[RY];
#pragma unroll
  for (int r = 0; r < RY; ++r) rin[r] = xin && (y0 + r < G.ny);

  const int64_t zbeg = (int64_t)zc * zchunk;
  int64_t zend = zbeg + zchunk;
  if (zend > G.nz) zend = G.nz;

  const T *pin_x = p_in;
  const T *pin_y = p_in + G.n;
  const T *pin_z = p_in + 2 * G.n;
  T *pout_x = p_out;
  T *pout_y = p_out + G.n;
  T *pout_z = p_out + 2 * G.n;

  // edge roles of this lane inside its wave patch
  const bool left_edge = (lx == 0);
  const bool right_edge = (lx == LX - 1);
  const bool top_edge = (ly == 0);
  const bool bottom_edge = (ly == LY - 1);
  const bool has_left = xin && left_edge && x0 > 0;
  const bool has_right = right_edge && (x0 + VEC < G.nx);
  const bool has_up = xin && top_edge && y0 > 0 && y0 - 1 < G.ny;
  const bool has_down = xin && bottom_edge && (y0 + RY < G.ny);

  T xc[RY][VEC];      // xbar[z]
  T pzprev[RY][VEC];  // new p_z at z-1

  int64_t off = zbeg * G.sz + y0 * G.sy + x0;  // (zbeg, y0, x0)
#pragma unroll
  for (int r = 0; r < RY; ++r) {
    zero(xc[r]);
    zero(pzprev[r]);
    if (rin[r]) ld(xbar_in + off + r * G.sy, xc[r]);
  }
  if constexpr (NDIM >= 3) {
    if (zbeg > 0) {
#pragma unroll
      for (int r = 0; r < RY; ++r) {
        if (rin[r]) {
          T xm[VEC], pm[VEC];
          zero(pm);
          ld(xbar_in + off - G.sz + r * G.sy, xm);
          if (S.has_p) ld(pin_z + off - G.sz + r * G.sy, pm);
#pragma unroll
          for (int k = 0; k < VEC; ++k)
            pzprev[r][k] = dual_update(pm[k], xc[r][k], xm[k], G.wz, S);
        }
      }
    }
  }

  for (int64_t z = zbeg; z < zend; ++z, off += G.sz) {
    // ---------------- loads of plane z (and xbar of plane z+1) ------------
    T xn[RY][VEC], xv[RY][VEC], bv[RY][VEC];
    T pxo[RY][VEC], pyo[RY][VEC], pzo[RY][VEC];
    const bool znext = (NDIM >= 3) && (z + 1 < G.nz);
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      zero(xn[r]); zero(xv[r]); zero(bv[r]);
      zero(pxo[r]); zero(pyo[r]); zero(pzo[r]);
      if (rin[r]) {
        const int64_t o = off + r * G.sy;
        if (znext) ld(xbar_in + o + G.sz, xn[r]);
        ld(x + o, xv[r]);
        ld(bt + o, bv[r]);
        if (S.has_p) {
          ld(pin_x + o, pxo[r]);
          if constexpr (NDIM >= 2) ld(pin_y + o, pyo[r]);
          if constexpr (NDIM >= 3) ld(pin_z + o, pzo[r]);
        }
      }
    }
    // halo: column to the right / left of the wave patch
    T xright[RY], xleft[RY], pxleft[RY];
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      xright[r] = T(0); xleft[r] = T(0); pxleft[r] = T(0);
      const int64_t o = off + r * G.sy;
      if (has_right && (y0 + r < G.ny)) xright[r] = xbar_in[o + VEC];
      if (has_left && rin[r]) {
        xleft[r] = xbar_in[o - 1];
        if (S.has_p) pxleft[r] = pin_x[o - 1];
      }
    }
    // halo: row above / below the wave patch
    T xdown[VEC], xup[VEC], pyup[VEC];
    zero(xdown); zero(xup); zero(pyup);
    if constexpr (NDIM >= 2) {
      if (has_down) ld(xbar_in + off + RY * G.sy, xdown);
      if (has_up) {
        ld(xbar_in + off - G.sy, xup);
        if (S.has_p) ld(pin_y + off - G.sy, pyup);
      }
    }

    // ---------------- dual update at the lane's own voxels ----------------
    T pxn[RY][VEC], pyn[RY][VEC], pzn[RY][VEC];
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      // x-neighbour to the right: next lane's first element
      T nb = __shfl_down(xc[r][0], 1, kWave);
      if (right_edge) nb = xright[r];
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const T hi = (k + 1 < VEC) ? xc[r][k + 1] : nb;
        pxn[r][k] = dual_update(pxo[r][k], hi, xc[r][k], G.wx, S);
      }
    }
    if constexpr (NDIM >= 2) {
      T below[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        if constexpr (LY > 1) below[k] = __shfl_down(xc[0][k], LX, kWave);
        else below[k] = T(0);
        if (bottom_edge) below[k] = xdown[k];
      }
#pragma unroll
      for (int r = 0; r < RY; ++r)
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const T hi = (r + 1 < RY) ? xc[(r + 1) % RY][k] : below[k];
          pyn[r][k] = dual_update(pyo[r][k], hi, xc[r][k], G.wy, S);
        }
    }
    if constexpr (NDIM >= 3) {
#pragma unroll
      for (int r = 0; r < RY; ++r)
#pragma unroll
        for (int k = 0; k < VEC; ++k)
          pzn[r][k] = dual_update(pzo[r][k], xn[r][k], xc[r][k], G.wz, S);
    }

    // ---------------- new dual values on the lower halo -------------------
    T pxl[RY];
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      pxl[r] = __shfl_up(pxn[r][VEC - 1], 1, kWave);
      if (left_edge)
        pxl[r] = has_left ? dual_update(pxleft[r], xc[r][0], xleft[r], G.wx, S)
                          : T(0);
    }
    T pyu[VEC];
    if constexpr (NDIM >= 2) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        if constexpr (LY > 1) pyu[k] = __shfl_up(pyn[RY - 1][k], LX, kWave);
        else pyu[k] = T(0);
        if (top_edge)
          pyu[k] = has_up ? dual_update(pyup[k], xc[0][k], xup[k], G.wy, S)
                          : T(0);
      }
    }

    // ---------------- primal update + stores ------------------------------
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      T xo_new[VEC], xb_new[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const T pl = (k > 0) ? pxn[r][(k + VEC - 1) % VEC] : pxl[r];
        T kt = pxn[r][k] * (-G.wx) + pl * G.wx;
        if constexpr (NDIM >= 2) {
          const T pu = (r > 0) ? pyn[(r + RY - 1) % RY][k] : pyu[k];
          kt += pyn[r][k] * (-G.wy) + pu * G.wy;
        }
        if constexpr (NDIM >= 3)
          kt += pzn[r][k] * (-G.wz) + pzprev[r][k] * G.wz;
        const T u = xv[r][k] - S.tau * kt;
        const T xnew = prox_data(u, bv[r][k], S.tl, S.one_plus_tl, S.l1 != 0);
        xo_new[k] = xnew;
        xb_new[k] = xnew + S.theta * (xnew - xv[r][k]);
      }
      if (rin[r]) {
        const int64_t o = off + r * G.sy;
        st(pout_x + o, pxn[r]);
        if constexpr (NDIM >= 2) st(pout_y + o, pyn[r]);
        if constexpr (NDIM >= 3) st(pout_z + o, pzn[r]);
        st(x + o, xo_new);
        st(xbar_out + o, xb_new);
      }
      if constexpr (NDIM >= 3) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          pzprev[r][k] = pzn[r][k];
          xc[r][k] = xn[r][k];
        }
      }
    }
  }
}

template <typename T, int VEC, int LX, int RY, int NDIM, bool RAG = false>
int launch_fused_t(const T *xbar_in, T *xbar_out, T *x, const T *bt,
                   const T *p_in, T *p_out, const Geom<T> &G,
                   const PdScalars<T> &S, hipStream_t st) {
  constexpr int LY = kWave / LX;
  constexpr int TY = (kBlock / kWave) * LY * RY;
  constexpr int TX = LX * VEC;
  const int64_t ntx = (G.nx + TX - 1) / TX;
  const int64_t nty = (G.ny + TY - 1) / TY;
  int64_t zchunk = g_tune.zchunk;
  if (zchunk <= 0) {
    // enough workgroups to fill 256 CUs several times; cache-resident volumes
    // get chunks as short as 2 planes (the extra plane per chunk is an L2 hit
    // there and 8 workgroups would leave the chip idle)
    const int64_t want = (4096 + ntx * nty - 1) / (ntx * nty);
    zchunk = (G.nz + want - 1) / want;
    if (zchunk < 2) zchunk = 2;
  }
  if (zchunk > G.nz) zchunk = G.nz;
  const int64_t nzc = (G.nz + zchunk - 1) / zchunk;
  int64_t slab = 0;
  int64_t blocks = ntx * nty * nzc;
  if (g_tune.xcd_map && nty >= 16) {
    slab = (nty + 7) / 8;
    blocks = 8 * slab * ntx * nzc;
  }
  if (blocks > 0x7fffffff) return NSOL_EINVAL;
  hipLaunchKernelGGL((k_pd_fused<T, VEC, LX, RY, NDIM, RAG>), dim3((unsigned)blocks),
                     dim3(kBlock), 0, st, xbar_in, xbar_out, x, bt, p_in, p_out,
                     G, S, (int)ntx, (int)nty, (int)zchunk, (int)slab);
  return launch_status();
}

template <typename T, int VEC, int LX, int RY, bool RAG = false>
int launch_fused_nd(const T *xbar_in, T *xbar_out, T *x, const T *bt,
                    const T *p_in, T *p_out, const Geom<T> &G,
                    const PdScalars<T> &S, hipStream_t st) {
  switch (G.ndim) {
    case 1: return launch_fused_t<T, VEC, LX, 1, 1, RAG>(xbar_in, xbar_out, x, bt, p_in, p_out, G, S, st);
    case 2: return launch_fused_t<T, VEC, LX, RY, 2, RAG>(xbar_in, xbar_out, x, bt, p_in, p_out, G, S, st);
    default: return launch_fused_t<T, VEC, LX, RY, 3, RAG>(xbar_in, xbar_out, x, bt, p_in, p_out, G, S, st);
  }
}

template <typename T, int VEC, int LX, bool RAG = false>
int launch_fused_ry(const T *xbar_in, T *xbar_out, T *x, const T *bt,
                    const T *p_in, T *p_out, const Geom<T> &G,
                    const PdScalars<T> &S, hipStream_t st) {
  int ry = g_tune.ry;
  if (ry == 0) {
    // two rows per lane unless that leaves fewer than ~2 workgroups per CU
    constexpr int TY2 = (kBlock / kWave) * (kWave / LX) * 2;
    const int64_t tiles = ((G.nx + LX * VEC - 1) / (LX * VEC)) * ((G.ny + TY2 - 1) / TY2);
    ry = (tiles * ((G.nz + 1) / 2) < 512) ? 1 : 2;
  }
  switch (ry) {
    case 1: return launch_fused_nd<T, VEC, LX, 1, RAG>(xbar_in, xbar_out, x, bt, p_in, p_out, G, S, st);
    case 4:
      if constexpr (!RAG)
        return launch_fused_nd<T, VEC, LX, 4>(xbar_in, xbar_out, x, bt, p_in, p_out, G, S, st);
    default: return launch_fused_nd<T, VEC, LX, 2, RAG>(xbar_in, xbar_out, x, bt, p_in, p_out, G, S, st);
  }
}

template <typename T>
inline bool aligned16(const T *a) {
  return (reinterpret_cast<uintptr_t>(a) & 15u) == 0;
}

template <typename T>
PdScalars<T> make_scalars(double sigma, double hden, double tau, double tl,
                          double theta, int flags, bool has_p) {
  PdScalars<T> S;
  S.sigma = (T)sigma; S.hden = huber_den<T>(hden); S.tau = (T)tau; S.tl = (T)tl;
  S.one_plus_tl = prox_den<T>(tl); S.theta = (T)theta;
  S.huber = (flags & NSOL_PD_REG_HUBER) ? 1 : 0;
  S.l1 = (flags & NSOL_PD_DATA_L1) ? 1 : 0;
  S.has_p = has_p ? 1 : 0;
  return S;
}

template <typename T>
int fused_iter_impl(const T *xbar_in, T *xbar_out, T *x, const T *bt,
                    const T *p_in, T *p_out, int ndim, int64_t nz, int64_t ny,
                    int64_t nx, double wx, double wy, double wz, double sigma,
                    double hden, double tau, double tl, double theta, int flags,
                    void *stream, int64_t pitch = 0) {
  NSOL_CHECK_GEOM(ndim, nz, ny, nx);
  if (!xbar_in || !xbar_out || !x || !bt || !p_out || xbar_in == xbar_out ||
      p_in == p_out)
    return NSOL_EINVAL;
  // (rows at a pitch: only the strides change -- the kernel below indexes rows and
  // planes by G.sy / G.sz and a gradient field's components by G.n)
  const Geom<T> G = make_geom_pitched<T>(ndim, nz, ny, nx, pitch, wx, wy, wz);
  const PdScalars<T> S =
      make_scalars<T>(sigma, hden, tau, tl, theta, flags, p_in != nullptr);
  hipStream_t st = as_stream(stream);
  constexpr int VW = 16 / sizeof(T);  // elements per 16-byte access
  const bool vec_ok = (nx % VW == 0) && aligned16(xbar_in) && aligned16(xbar_out) &&
                      aligned16(x) && aligned16(bt) && aligned16(p_out) &&
                      (!p_in || aligned16(p_in)) && ((nz * ny * nx) % VW == 0);
  if (G.padded && !(g_tune.rag && nx >= 2 * VW) && !vec_ok) return NSOL_EINVAL;
  if (vec_ok) {
    if (nx / VW >= kWave)
      return launch_fused_ry<T, VW, 64>(xbar_in, xbar_out, x, bt, p_in, p_out, G, S, st);
    return launch_fused_ry<T, VW, 16>(xbar_in, xbar_out, x, bt, p_in, p_out, G, S, st);
  }
  // rows that are not a multiple of 16 bytes / unaligned arrays: element-aligned
  // 16-byte accesses (g_tune.rag = 0 restores the 4-byte form, for the tests)
  if (g_tune.rag && nx >= 2 * VW) {
    if ((nx + VW - 1) / VW >= kWave)
      return launch_fused_ry<T, VW, 64, true>(xbar_in, xbar_out, x, bt, p_in, p_out, G, S, st);
    return launch_fused_ry<T, VW, 16, true>(xbar_in, xbar_out, x, bt, p_in, p_out, G, S, st);
  }
  if (nx >= kWave)
    return launch_fused_ry<T, 1, 64>(xbar_in, xbar_out, x, bt, p_in, p_out, G, S, st);
  return launch_fused_ry<T, 1, 16>(xbar_in, xbar_out, x, bt, p_in, p_out, G, S, st);
}

template <typename T>
int dual_step_impl(const T *xbar, const T *p_in, T *p_out, int ndim, int64_t nz,
                   int64_t ny, int64_t nx, double wx, double wy, double wz,
                   double sigma, double hden, void *stream) {
  NSOL_CHECK_GEOM(ndim, nz, ny, nx);
  if (!xbar || !p_out) return NSOL_EINVAL;
  const Geom<T> G = make_geom<T>(ndim, nz, ny, nx, wx, wy, wz);
  hipLaunchKernelGGL(k_dual_step<T>, dim3(grid_for(G.n)), dim3(kBlock), 0,
                     as_stream(stream), xbar, p_in, p_out, G, (T)sigma, huber_den<T>(hden),
                     hden != 1.0);
  return launch_status();
}

template <typename T>
int primal_step_impl(const T *p, T *x, T *xbar, const T *bt, int ndim,
                     int64_t nz, int64_t ny, int64_t nx, double wx, double wy,
                     double wz, double tau, double tl, double theta, int flags,
                     void *stream) {
  NSOL_CHECK_GEOM(ndim, nz, ny, nx);
  if (!p || !x || !xbar || !bt) return NSOL_EINVAL;
  const Geom<T> G = make_geom<T>(ndim, nz, ny, nx, wx, wy, wz);
  hipLaunchKernelGGL(k_primal_step<T>, dim3(grid_for(G.n)), dim3(kBlock), 0,
                     as_stream(stream), p, x, xbar, bt, G, (T)tau, (T)tl,
                     prox_den<T>(tl), (T)theta, (flags & NSOL_PD_DATA_L1) != 0);
  return launch_status();
}

inline int fused2_call(const float *a, float *b, const float *c, float *d,
                       const float *e, const float *f, float *g, int ndim,
                       int64_t nz, int64_t ny, int64_t nx, double wx, double wy,
                       double wz, const double *s, const double *h,
                       const double *t, const double *tl, const double *th,
                       int flags, void *st) {
  return nsol_pd_fused2_iter_f32(a, b, c, d, e, f, g, ndim, nz, ny, nx, wx, wy, wz,
                                 s, h, t, tl, th, flags, st);
}
inline int fused2_call(const double *a, double *b, const double *c, double *d,
                       const double *e, const double *f, double *g, int ndim,
                       int64_t nz, int64_t ny, int64_t nx, double wx, double wy,
                       double wz, const double *s, const double *h,
                       const double *t, const double *tl, const double *th,
                       int flags, void *st) {
  return nsol_pd_fused2_iter_f64(a, b, c, d, e, f, g, ndim, nz, ny, nx, wx, wy, wz,
                                 s, h, t, tl, th, flags, st);
}

inline int fusedk_call(const float *a, float *b, const float *c, float *d,
                       const float *e, const float *f, float *g, int ndim,
                       int64_t nz, int64_t ny, int64_t nx, double wx, double wy,
                       double wz, int k, const double *s, const double *h,
                       const double *t, const double *tl, const double *th,
                       int flags, void *st) {
  return nsol_pd_fusedk_iter_f32(a, b, c, d, e, f, g, ndim, nz, ny, nx, wx, wy, wz,
                                 k, s, h, t, tl, th, flags, st);
}
inline int fusedk_call(const double *a, double *b, const double *c, double *d,
                       const double *e, const double *f, double *g, int ndim,
                       int64_t nz, int64_t ny, int64_t nx, double wx, double wy,
                       double wz, int k, const double *s, const double *h,
                       const double *t, const double *tl, const double *th,
                       int flags, void *st) {
  return nsol_pd_fusedk_iter_f64(a, b, c, d, e, f, g, ndim, nz, ny, nx, wx, wy, wz,
                                 k, s, h, t, tl, th, flags, st);
}

extern "C" int nsol_pd_fusedk_tail2(int elem_size, int64_t nz, int64_t ny, int64_t nx);

extern "C" int nsol_pd_fusedk_tail2_pitched(int elem_size, int64_t nz, int64_t ny, int64_t nx,
                                            int64_t pitch);
extern "C" int nsol_pd_fusedk_iter_pitched_f32(
    const float *, float *, const float *, float *, const float *, const float *, float *, int,
    int64_t, int64_t, int64_t, int64_t, double, double, double, int, const double *,
    const double *, const double *, const double *, const double *, int, void *);
extern "C" int nsol_pd_fusedk_iter_pitched_f64(
    const double *, double *, const double *, double *, const double *, const double *,
    double *, int, int64_t, int64_t, int64_t, int64_t, double, double, double, int,
    const double *, const double *, const double *, const double *, const double *, int,
    void *);
inline int fusedk_pitched(const float *a, float *b, const float *c, float *d, const float *e,
                          const float *f, float *g, int ndim, int64_t nz, int64_t ny,
                          int64_t nx, int64_t pitch, double wx, double wy, double wz, int k,
                          const double *s, const double *h, const double *t, const double *tl,
                          const double *th, int flags, void *st) {
  return nsol_pd_fusedk_iter_pitched_f32(a, b, c, d, e, f, g, ndim, nz, ny, nx, pitch, wx, wy,
                                         wz, k, s, h, t, tl, th, flags, st);
}
inline int fusedk_pitched(const double *a, double *b, const double *c, double *d,
                          const double *e, const double *f, double *g, int ndim, int64_t nz,
                          int64_t ny, int64_t nx, int64_t pitch, double wx, double wy,
                          double wz, int k, const double *s, const double *h, const double *t,
                          const double *tl, const double *th, int flags, void *st) {
  return nsol_pd_fusedk_iter_pitched_f64(a, b, c, d, e, f, g, ndim, nz, ny, nx, pitch, wx, wy,
                                         wz, k, s, h, t, tl, th, flags, st);
}

// pitch > nx: every array holds its rows at that pitch (elements; whole 16-byte vectors),
// planes ny * pitch apart, the components of p nz * ny * pitch apart -- the layout
// nsol_amd's solver keeps volumes in whose rows are not whole vectors: aligned accesses,
// the row's partial vector masked in registers, the padding free to hold anything.
template <typename T>
int run_impl(T *xbar0, T *xbar1, T *x, T *x_alt, const T *bt, T *p0, T *p1,
             int ndim, int64_t nz, int64_t ny, int64_t nx, double wx, double wy,
             double wz, double lambda, const double *sig, const double *tau,
             const double *theta, int iterations, int p_is_zero,
             double gamma_huber, int flags, int *final_slot, void *stream,
             int64_t pitch = 0) {
  NSOL_CHECK_GEOM(ndim, nz, ny, nx);
  if (iterations < 0 || !sig || !tau || !theta || !x) return NSOL_EINVAL;
  const bool pitched = pitch > nx;
  if (pitch > 0 && (pitch < nx || ndim != 3 || pitch % (16 / (int64_t)sizeof(T)) != 0))
    return NSOL_EINVAL;
  const bool may_swap = (flags & NSOL_PD_RUN_X_MAY_SWAP) != 0 && final_slot != nullptr;
  flags &= ~NSOL_PD_RUN_X_MAY_SWAP;
  T *xb[2] = {xbar0, xbar1};
  T *pp[2] = {p0, p1};
  T *xcur = x, *xoth = x_alt;
  int slot = 0;
  const bool huber = (flags & NSOL_PD_REG_HUBER) != 0;
  int n = 0;
  while (n < iterations) {
    const T *pin = (n == 0 && p_is_zero) ? nullptr : pp[slot];
    int rc = -2;
    if (xoth && n + 1 < iterations && !g_tune.force_two_pass) {
      // deepest temporal blocking first: 3 iterations per pass (tiled
      // footprints), then 2 (tiled or full-row footprints), then 1
      double h3[3], tl3[3];
      const int left = iterations - n < 3 ? iterations - n : 3;
      for (int i = 0; i < left; ++i) {
        h3[i] = huber ? 1.0 + sig[n + i] * gamma_huber : 1.0;
        tl3[i] = tau[n + i] * lambda;
      }
      int done = 0;
      if (pitched) {
        // rows at a pitch: depth 3, then depth 2 of k_pd_fusedk (k_pd_fused2 wants
        // contiguous whole rows), then one iteration at a time
        for (int depth = left >= 3 ? 3 : 2; depth >= 2 && !done; --depth) {
          rc = fusedk_pitched(xb[slot], xb[slot ^ 1], xcur, xoth, bt, pin, pp[slot ^ 1],
                              ndim, nz, ny, nx, pitch, wx, wy, wz, depth, sig + n, h3,
                              tau + n, tl3, theta + n, flags, stream);
          if (rc == 0) done = depth;
          else if (rc != -2) return rc;
        }
        if (done) {
          n += done;
          slot ^= 1;
          T *t = xcur; xcur = xoth; xoth = t;
          continue;
        }
        rc = -2;
      }
      if (!pitched && left == 2 && nsol_pd_fusedk_tail2((int)sizeof(T), nz, ny, nx)) {
        // trailing pair of a run on a shape whose depth-3 plan has settled
        rc = fusedk_call(xb[slot], xb[slot ^ 1], xcur, xoth, bt, pin, pp[slot ^ 1],
                         ndim, nz, ny, nx, wx, wy, wz, 2, sig + n, h3, tau + n, tl3,
                         theta + n, flags, stream);
        if (rc == 0) done = 2;
        else if (rc != -2) return rc;
      }
      if (!pitched && !done && left >= 3) {
        rc = fusedk_call(xb[slot], xb[slot ^ 1], xcur, xoth, bt, pin, pp[slot ^ 1],
                         ndim, nz, ny, nx, wx, wy, wz, 3, sig + n, h3, tau + n, tl3,
                         theta + n, flags, stream);
        if (rc == 0) done = 3;
        else if (rc != -2) return rc;
      }
      if (!pitched && !done) {
        rc = fused2_call(xb[slot], xb[slot ^ 1], xcur, xoth, bt, pin, pp[slot ^ 1],
                         ndim, nz, ny, nx, wx, wy, wz, sig + n, h3, tau + n, tl3,
                         theta + n, flags, stream);
        if (rc == 0) done = 2;
        else if (rc != -2) return rc;
      }
      if (!pitched && !done) {   // e.g. rows too short for the full-row footprints
        rc = fusedk_call(xb[slot], xb[slot ^ 1], xcur, xoth, bt, pin, pp[slot ^ 1],
                         ndim, nz, ny, nx, wx, wy, wz, 2, sig + n, h3, tau + n, tl3,
                         theta + n, flags, stream);
        if (rc == 0) done = 2;
        else if (rc != -2) return rc;
      }
      if (done) {
        n += done;
        slot ^= 1;
        T *t = xcur; xcur = xoth; xoth = t;
        continue;
      }
    }
    const double hden = huber ? 1.0 + sig[n] * gamma_huber : 1.0;
    if (g_tune.force_two_pass && !pitched) {
      rc = dual_step_impl<T>(xb[slot], pin, pp[slot ^ 1], ndim, nz, ny, nx, wx, wy,
                             wz, sig[n], hden, stream);
      if (rc) return rc;
      rc = primal_step_impl<T>(pp[slot ^ 1], xcur, xb[slot ^ 1], bt, ndim, nz, ny,
                               nx, wx, wy, wz, tau[n], tau[n] * lambda, theta[n],
                               flags, stream);
    } else {
      rc = fused_iter_impl<T>(xb[slot], xb[slot ^ 1], xcur, bt, pin, pp[slot ^ 1],
                              ndim, nz, ny, nx, wx, wy, wz, sig[n], hden, tau[n],
                              tau[n] * lambda, theta[n], flags, stream, pitch);
    }
    if (rc) return rc;
    n += 1;
    slot ^= 1;
  }
  if (xcur != x && !may_swap) {
    hipError_t e = hipMemcpyAsync(x, xcur,
                                  sizeof(T) * (size_t)(nz * ny * (pitched ? pitch : nx)),
                                  hipMemcpyDeviceToDevice, as_stream(stream));
    if (e != hipSuccess) return (int)e;
  }
  if (final_slot) *final_slot = slot | ((xcur != x && may_swap) ? 2 : 0);
  return 0;
}

}  // namespace

extern "C" {

/* tuning knobs for experiments: "pd_zchunk", "pd_ry", "pd_two_pass" */
int nsol_hip_set_param(const char *name, int value) {
  if (!name) return NSOL_EINVAL;
  if (!strcmp(name, "pd_zchunk")) g_tune.zchunk = value;
  else if (!strcmp(name, "pd_ry")) g_tune.ry = value;
  else if (!strcmp(name, "pd_two_pass")) g_tune.force_two_pass = value;
  else if (!strcmp(name, "pd_xcd_map")) g_tune.xcd_map = value;
  else if (!strcmp(name, "pd_rag")) g_tune.rag = value;
  else if (!strcmp(name, "stencil_slabs")) g_stencil_slabs = value ? 1 : 0;
  else if (!strcmp(name, "stencil_blocks"))
    g_stencil_blocks = value < 256 ? 256 : (value > kReducePartials ? kReducePartials : value);
  else if (!strcmp(name, "max_grid_blocks"))
    g_max_grid_blocks = value < 1 ? 1 : (value > kMaxGridBlocksLimit ? kMaxGridBlocksLimit : value);
  else return NSOL_EINVAL;
  return 0;
}

#define NSOL_PD_DEF(T, SUF)                                                      \
  int nsol_pd_dual_step_##SUF(const T *xbar, const T *p_in, T *p_out, int ndim,  \
                              int64_t nz, int64_t ny, int64_t nx, double wx,     \
                              double wy, double wz, double sigma, double hden,   \
                              void *s) {                                         \
    return dual_step_impl<T>(xbar, p_in, p_out, ndim, nz, ny, nx, wx, wy, wz,    \
                             sigma, hden, s);                                    \
  }                                                                              \
  int nsol_pd_primal_step_##SUF(const T *p, T *x, T *xbar, const T *bt,          \
                                int ndim, int64_t nz, int64_t ny, int64_t nx,    \
                                double wx, double wy, double wz, double tau,     \
                                double tl, double theta, int flags, void *s) {   \
    return primal_step_impl<T>(p, x, xbar, bt, ndim, nz, ny, nx, wx, wy, wz,     \
                               tau, tl, theta, flags, s);                        \
  }                                                                              \
  int nsol_pd_fused_iter_##SUF(const T *xi, T *xo, T *x, const T *bt,            \
                               const T *pi, T *po, int ndim, int64_t nz,         \
                               int64_t ny, int64_t nx, double wx, double wy,     \
                               double wz, double sigma, double hden, double tau, \
                               double tl, double theta, int flags, void *s) {    \
    return fused_iter_impl<T>(xi, xo, x, bt, pi, po, ndim, nz, ny, nx, wx, wy,   \
                              wz, sigma, hden, tau, tl, theta, flags, s);        \
  }                                                                              \
  int nsol_pd_run_##SUF(T *xb0, T *xb1, T *x, T *x_alt, const T *bt, T *p0,      \
                        T *p1, int ndim, int64_t nz, int64_t ny, int64_t nx,     \
                        double wx, double wy, double wz, double lambda,          \
                        const double *sg, const double *ta, const double *th,    \
                        int iters, int p_is_zero, double gh, int flags,          \
                        int *final_slot, void *s) {                              \
    return run_impl<T>(xb0, xb1, x, x_alt, bt, p0, p1, ndim, nz, ny, nx, wx, wy,  \
                       wz, lambda, sg, ta, th, iters, p_is_zero, gh, flags,      \
                       final_slot, s);                                           \
  }

NSOL_PD_DEF(float, f32)
NSOL_PD_DEF(double, f64)

#define NSOL_PD_PITCHED(T, SUF)                                                   \
  int nsol_pd_run_pitched_##SUF(T *xb0, T *xb1, T *x, T *x_alt, const T *bt, T *p0, \
                                T *p1, int ndim, int64_t nz, int64_t ny, int64_t nx, \
                                int64_t pitch, double wx, double wy, double wz,    \
                                double lambda, const double *sg, const double *ta, \
                                const double *th, int iters, int p_is_zero,        \
                                double gh, int flags, int *final_slot, void *s) {  \
    return run_impl<T>(xb0, xb1, x, x_alt, bt, p0, p1, ndim, nz, ny, nx, wx, wy,   \
                       wz, lambda, sg, ta, th, iters, p_is_zero, gh, flags,        \
                       final_slot, s, pitch);                                      \
  }
NSOL_PD_PITCHED(float, f32)
NSOL_PD_PITCHED(double, f64)
#undef NSOL_PD_PITCHED

}  // extern "C"
