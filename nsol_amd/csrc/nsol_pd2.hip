// Two Chambolle-Pock iterations in ONE pass over memory (temporal blocking).
//
// reference: two consecutive trips of the loop body primal_dual_solver.py:242-256.
//
// Per pair of iterations the kernel reads xbar, x, bt, p[3] once and writes
// p[3], x, xbar once: 11 words per voxel for TWO iterations (the one-iteration
// kernel k_pd_fused moves 11 words per iteration).
//
// Scheme.  A workgroup of WX x WY waves owns a footprint of WX*64*VEC voxels in
// x by WY*RY rows and marches along z.  At step s it
//   (1) runs iteration n+1 on plane s exactly like k_pd_fused (old-data halos
//       come from L1/L2), keeping x', xbar', p' in registers only;
//   (2) finishes iteration n+2 on plane s-1, which was waiting for xbar'[s]
//       (z-neighbour): p''_z, K^T p'', prox, over-relaxation, stores;
//   (3) starts iteration n+2 on plane s: the waves publish the edge rows /
//       columns of xbar', p'_x, p'_y in LDS (one barrier per plane, double
//       buffered), x-neighbours inside a wave travel by shuffles, and p''_x,
//       p''_y plus the in-plane part of K^T p'' are formed and stored.
// Iteration n+2 needs iteration-n+1 values one voxel beyond its own region, so
// footprints overlap by one row above/below (and one vector left/right when
// the volume is wider than the footprint): those voxels are recomputed, never
// exchanged between workgroups, and the second-iteration results are stored
// only for the interior.  Along z the pipeline skew makes the overlap one plane
// per chunk end.  Arithmetic per voxel is identical to two launches of
// k_pd_fused (same operation order), so results are bit-identical.
#include <string.h>

#include "nsol_common.hpp"
#include "nsol_pd_common.hpp"

using namespace nsol;

namespace nsol_pd2 {

// LDS image one wave publishes per plane
template <typename T, int VEC, int RY>
struct WaveEdges {
  T row_top[64 * VEC];     // xbar' of the wave's first row
  T row_bot[64 * VEC];     // xbar' of its last row
  T row_bot_py[64 * VEC];  // p'_y of its last row
  T col_l[RY];             // xbar' at the first voxel of each row (lane 0)
  T col_r[RY];             // xbar' at the last voxel of each row (lane 63)
  T col_r_px[RY];          // p'_x there
  T pad[(16 / sizeof(T)) * 2 - (3 * RY) % ((16 / sizeof(T)) * 2) ];
};

template <typename T, int VEC, int WX, int WY, int RY, int WPE, bool HUBER, bool L1>
__global__ __launch_bounds__(WX *WY * 64, WPE) void k_pd_fused2(
    const T *__restrict__ xbar_in, T *__restrict__ xbar_out,
    const T *__restrict__ x_in, T *__restrict__ x_out,
    const T *__restrict__ bt, const T *__restrict__ p_in,
    T *__restrict__ p_out, Geom<T> G, PdScalars<T> S1, PdScalars<T> S2, int ntx,
    int nty, int zchunk, int slab) {
  constexpr int NW = WX * WY;
  constexpr int TXB = WX * 64 * VEC;   // footprint in x
  constexpr int TYV = WY * RY - 2;     // rows with valid second-iteration output
  typedef WaveEdges<T, VEC, RY> Edges;
  __shared__ __attribute__((aligned(16))) Edges lds[2][NW];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wxi = wave % WX;
  const int wyi = wave / WX;
  // Block -> footprint map.  blockIdx % 8 names the XCD (shared L2) under the
  // round-robin dispatch; with slab > 0 each XCD walks `slab` consecutive
  // y-tiles, so the rows that overlapping footprints both read meet in one L2.
  // The whole workgroup takes the same branch (no barrier is skipped).
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

  // ---- geometry of this lane
  const bool single_x = (ntx == 1);
  const int64_t XV = TXB - 2 * VEC;
  const int64_t xv_lo = single_x ? 0 : (int64_t)tx * XV;   // valid x range
  const int64_t xv_hi = single_x ? G.nx : xv_lo + XV;
  const int64_t xb0 = single_x ? 0 : xv_lo - VEC;
  const int64_t x0 = xb0 + (int64_t)(wxi * 64 + lane) * VEC;
  const int64_t yv_lo = (int64_t)ty * TYV;
  const int64_t yv_hi = yv_lo + TYV;
  const int64_t y0 = yv_lo - 1 + (int64_t)wyi * RY;
  const bool xin = x0 >= 0 && x0 < G.nx;
  const bool xvalid = xin && x0 >= xv_lo && x0 < xv_hi;
  bool rin[RY], rvalid[RY];
#pragma unroll
  for (int r = 0; r < RY; ++r) {
    const int64_t y = y0 + r;
    rin[r] = xin && y >= 0 && y < G.ny;
    rvalid[r] = rin[r] && xvalid && y >= yv_lo && y < yv_hi;
  }
  // Step sizes masked to zero outside the volume: a voxel there then gets
  // p' = clamp(0 + 0*g) = 0 and x' = prox(0 - 0*kt) = 0 exactly (K and K^T pad
  // with zeros) without per-value selects; inside the volume nothing changes.
  T sig1[RY], sig2[RY], tau1[RY];
#pragma unroll
  for (int r = 0; r < RY; ++r) {
    sig1[r] = rin[r] ? S1.sigma : T(0);
    sig2[r] = rin[r] ? S2.sigma : T(0);
    tau1[r] = rin[r] ? S1.tau : T(0);
  }
  const int64_t zbeg = (int64_t)zc * zchunk;
  int64_t zend = zbeg + zchunk;
  if (zend > G.nz) zend = G.nz;

  const T *pin_x = p_in, *pin_y = p_in + G.n, *pin_z = p_in + 2 * G.n;
  T *pout_x = p_out, *pout_y = p_out + G.n, *pout_z = p_out + 2 * G.n;

  // halo roles for iteration n+1 (old data, global memory)
  const bool has_left = xin && lane == 0 && x0 > 0;
  const bool has_right = lane == 63 && x0 >= 0 && (x0 + VEC < G.nx);
  const bool has_up = xin && (y0 - 1 >= 0) && (y0 - 1 < G.ny);
  const bool has_down = xin && (y0 + RY >= 0) && (y0 + RY < G.ny);
  // neighbours for iteration n+2 (new data, LDS / shuffles)
  const bool nb_right = wxi + 1 < WX;
  const bool nb_left = wxi > 0;
  const bool nb_down = wyi + 1 < WY;
  const bool nb_up = wyi > 0;
  const bool dom_up = (y0 - 1 >= 0);     // row above the wave is inside the volume
  const bool dom_left = (x0 - 1 >= 0);

  const int64_t s_first = zbeg > 0 ? zbeg - 1 : 0;
  int64_t off = s_first * G.sz + y0 * G.sy + x0;

  T xc[RY][VEC];       // xbar[s]
  T p1z[RY][VEC];      // p'_z[s-1] (new dual of iteration n+1)
  T c_xb1[RY][VEC];    // xbar'[s-1]
  T c_x1[RY][VEC];     // x'[s-1]
  T c_bt[RY][VEC];     // bt[s-1]
  T c_kt[RY][VEC];     // in-plane part of K^T p'' at s-1
  T p2z_prev[RY][VEC]; // p''_z[s-2]
#pragma unroll
  for (int r = 0; r < RY; ++r) {
    zero(xc[r]); zero(p1z[r]); zero(c_xb1[r]); zero(c_x1[r]); zero(c_bt[r]);
    zero(c_kt[r]); zero(p2z_prev[r]);
    if (rin[r]) ldv<T, VEC>(xbar_in + off + r * G.sy, xc[r]);
  }
  if (s_first > 0) {
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      if (rin[r]) {
        T xm[VEC], pm[VEC];
        zero(pm);
        ldv<T, VEC>(xbar_in + off - G.sz + r * G.sy, xm);
        if (S1.has_p) ldv<T, VEC>(pin_z + off - G.sz + r * G.sy, pm);
#pragma unroll
        for (int k = 0; k < VEC; ++k)
          p1z[r][k] = dual_update_s<HUBER>(pm[k], xc[r][k], xm[k], G.wz, S1.sigma,
                                           S1.hden);
      }
    }
  }

  for (int64_t s = s_first; s <= zend; ++s, off += G.sz) {
    const bool have1 = s < G.nz;
    T xb1[RY][VEC], x1[RY][VEC], btc[RY][VEC];
    T p1x[RY][VEC], p1y[RY][VEC], p1zn[RY][VEC];
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      zero(xb1[r]); zero(x1[r]); zero(btc[r]);
      zero(p1x[r]); zero(p1y[r]); zero(p1zn[r]);
    }
    if (have1) {
      // ================= iteration n+1 on plane s (as k_pd_fused) ==========
      T xn[RY][VEC], xv[RY][VEC], pxo[RY][VEC], pyo[RY][VEC], pzo[RY][VEC];
      const bool znext = (s + 1 < G.nz);
#pragma unroll
      for (int r = 0; r < RY; ++r) {
        zero(xn[r]); zero(xv[r]); zero(pxo[r]); zero(pyo[r]); zero(pzo[r]);
        if (rin[r]) {
          const int64_t o = off + r * G.sy;
          if (znext) ldv<T, VEC>(xbar_in + o + G.sz, xn[r]);
          ldv<T, VEC>(x_in + o, xv[r]);
          ldv<T, VEC>(bt + o, btc[r]);
          if (S1.has_p) {
            ldv<T, VEC>(pin_x + o, pxo[r]);
            ldv<T, VEC>(pin_y + o, pyo[r]);
            ldv<T, VEC>(pin_z + o, pzo[r]);
          }
        }
      }
      T xright[RY], xleft[RY], pxleft[RY];
#pragma unroll
      for (int r = 0; r < RY; ++r) {
        xright[r] = T(0); xleft[r] = T(0); pxleft[r] = T(0);
        const int64_t y = y0 + r;
        const int64_t o = off + r * G.sy;
        if (has_right && y >= 0 && y < G.ny) xright[r] = xbar_in[o + VEC];
        if (has_left && rin[r]) {
          xleft[r] = xbar_in[o - 1];
          if (S1.has_p) pxleft[r] = pin_x[o - 1];
        }
      }
      T xdown[VEC], xup[VEC], pyup[VEC];
      zero(xdown); zero(xup); zero(pyup);
      if (has_down) ldv<T, VEC>(xbar_in + off + RY * G.sy, xdown);
      if (has_up) {
        ldv<T, VEC>(xbar_in + off - G.sy, xup);
        if (S1.has_p) ldv<T, VEC>(pin_y + off - G.sy, pyup);
      }
      // dual update (zero outside the volume: K^T pads with zeros)
#pragma unroll
      for (int r = 0; r < RY; ++r) {
        T nb = __shfl_down(xc[r][0], 1, kWave);
        if (lane == 63) nb = xright[r];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const T hx = (k + 1 < VEC) ? xc[r][(k + 1) % VEC] : nb;
          const T hy = (r + 1 < RY) ? xc[(r + 1) % RY][k] : xdown[k];
          p1x[r][k] = dual_update_s<HUBER>(pxo[r][k], hx, xc[r][k], G.wx, sig1[r], S1.hden);
          p1y[r][k] = dual_update_s<HUBER>(pyo[r][k], hy, xc[r][k], G.wy, sig1[r], S1.hden);
          p1zn[r][k] = dual_update_s<HUBER>(pzo[r][k], xn[r][k], xc[r][k], G.wz, sig1[r], S1.hden);
        }
      }
      T pxl[RY], pyu[VEC];
#pragma unroll
      for (int r = 0; r < RY; ++r) {
        pxl[r] = __shfl_up(p1x[r][VEC - 1], 1, kWave);
        if (lane == 0)
          pxl[r] = (has_left && rin[r])
                       ? dual_update_s<HUBER>(pxleft[r], xc[r][0], xleft[r], G.wx,
                                              S1.sigma, S1.hden)
                       : T(0);
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k)
        pyu[k] = has_up ? dual_update_s<HUBER>(pyup[k], xc[0][k], xup[k], G.wy,
                                               S1.sigma, S1.hden)
                        : T(0);
      // primal update, kept in registers
#pragma unroll
      for (int r = 0; r < RY; ++r)
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const T pl = (k > 0) ? p1x[r][(k + VEC - 1) % VEC] : pxl[r];
          const T pu = (r > 0) ? p1y[(r + RY - 1) % RY][k] : pyu[k];
          T kt = p1x[r][k] * (-G.wx) + pl * G.wx;
          kt += p1y[r][k] * (-G.wy) + pu * G.wy;
          kt += p1zn[r][k] * (-G.wz) + p1z[r][k] * G.wz;
          const T u = xv[r][k] - tau1[r] * kt;
          const T xnew = prox_data_s<L1>(u, btc[r][k], S1.tl, S1.one_plus_tl);
          x1[r][k] = xnew;
          xb1[r][k] = xnew + S1.theta * (xnew - xv[r][k]);
        }
#pragma unroll
      for (int r = 0; r < RY; ++r)
#pragma unroll
        for (int k = 0; k < VEC; ++k) xc[r][k] = xn[r][k];
    }

    // ================= finish iteration n+2 on plane s-1 ====================
    if (s > s_first) {
      const bool own = (s - 1 >= zbeg);
#pragma unroll
      for (int r = 0; r < RY; ++r) {
        T p2z[VEC], x2[VEC], xb2[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          p2z[k] = dual_update_s<HUBER>(p1z[r][k], xb1[r][k], c_xb1[r][k], G.wz,
                                        sig2[r], S2.hden);
          T kt = c_kt[r][k];
          kt += p2z[k] * (-G.wz) + p2z_prev[r][k] * G.wz;
          const T u = c_x1[r][k] - S2.tau * kt;
          x2[k] = prox_data_s<L1>(u, c_bt[r][k], S2.tl, S2.one_plus_tl);
          xb2[k] = x2[k] + S2.theta * (x2[k] - c_x1[r][k]);
          p2z_prev[r][k] = p2z[k];
        }
        if (own && rvalid[r]) {
          const int64_t o = off - G.sz + r * G.sy;
          stv<T, VEC>(pout_z + o, p2z);
          stv<T, VEC>(x_out + o, x2);
          stv<T, VEC>(xbar_out + o, xb2);
        }
      }
    }

    // ================= in-plane part of iteration n+2 on plane s ============
    if (s >= zbeg && s < zend) {
      Edges &mine = lds[s & 1][wave];
      stv<T, VEC>(&mine.row_top[lane * VEC], xb1[0]);
      stv<T, VEC>(&mine.row_bot[lane * VEC], xb1[RY - 1]);
      stv<T, VEC>(&mine.row_bot_py[lane * VEC], p1y[RY - 1]);
      if (lane == 0) {
#pragma unroll
        for (int r = 0; r < RY; ++r) mine.col_l[r] = xb1[r][0];
      }
      if (lane == 63) {
#pragma unroll
        for (int r = 0; r < RY; ++r) {
          mine.col_r[r] = xb1[r][VEC - 1];
          mine.col_r_px[r] = p1x[r][VEC - 1];
        }
      }
      __syncthreads();
      T below[VEC], above[VEC], above_py[VEC];
      zero(below); zero(above); zero(above_py);
      if (nb_down) ldv<T, VEC>(&lds[s & 1][wave + WX].row_top[lane * VEC], below);
      if (nb_up) {
        ldv<T, VEC>(&lds[s & 1][wave - WX].row_bot[lane * VEC], above);
        ldv<T, VEC>(&lds[s & 1][wave - WX].row_bot_py[lane * VEC], above_py);
      }
      T p2x[RY][VEC], p2y[RY][VEC];
#pragma unroll
      for (int r = 0; r < RY; ++r) {
        T nb = __shfl_down(xb1[r][0], 1, kWave);
        if (lane == 63) nb = nb_right ? lds[s & 1][wave + 1].col_l[r] : T(0);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const T hx = (k + 1 < VEC) ? xb1[r][(k + 1) % VEC] : nb;
          const T hy = (r + 1 < RY) ? xb1[(r + 1) % RY][k] : below[k];
          p2x[r][k] = dual_update_s<HUBER>(p1x[r][k], hx, xb1[r][k], G.wx, sig2[r], S2.hden);
          p2y[r][k] = dual_update_s<HUBER>(p1y[r][k], hy, xb1[r][k], G.wy, sig2[r], S2.hden);
        }
      }
      T p2xl[RY], p2yu[VEC];
#pragma unroll
      for (int r = 0; r < RY; ++r) {
        p2xl[r] = __shfl_up(p2x[r][VEC - 1], 1, kWave);
        if (lane == 0) {
          p2xl[r] = T(0);
          if (nb_left && dom_left && rin[r])
            p2xl[r] = dual_update_s<HUBER>(lds[s & 1][wave - 1].col_r_px[r],
                                           xb1[r][0], lds[s & 1][wave - 1].col_r[r],
                                           G.wx, S2.sigma, S2.hden);
        }
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k)
        p2yu[k] = (nb_up && dom_up && xin)
                      ? dual_update_s<HUBER>(above_py[k], xb1[0][k], above[k], G.wy,
                                             S2.sigma, S2.hden)
                      : T(0);
#pragma unroll
      for (int r = 0; r < RY; ++r) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const T pl = (k > 0) ? p2x[r][(k + VEC - 1) % VEC] : p2xl[r];
          const T pu = (r > 0) ? p2y[(r + RY - 1) % RY][k] : p2yu[k];
          T kt = p2x[r][k] * (-G.wx) + pl * G.wx;
          kt += p2y[r][k] * (-G.wy) + pu * G.wy;
          c_kt[r][k] = kt;
        }
        if (rvalid[r]) {
          const int64_t o = off + r * G.sy;
          stv<T, VEC>(pout_x + o, p2x[r]);
          stv<T, VEC>(pout_y + o, p2y[r]);
        }
      }
    }

    // ================= carry plane s =====================================
#pragma unroll
    for (int r = 0; r < RY; ++r)
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        c_xb1[r][k] = xb1[r][k];
        c_x1[r][k] = x1[r][k];
        c_bt[r][k] = btc[r][k];
        p1z[r][k] = p1zn[r][k];
      }
  }
}

struct Tuning {
  int zchunk = 0;
  int enable = 1;
  int xcd_map = 1;
  int variant = 0;   // 0 = one row per lane, (WX x 8) waves [default: 114 VGPRs,
                     //     4 waves/SIMD, measured 0.79 ms/iteration at 512^3];
                     // 1 = two rows per lane, (WX x 4) waves [203 VGPRs, 0.98 ms]
};

inline int cu_count() {
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

// z-chunk length: trade the one-plane overlap per chunk end against filling
// the last round of workgroups (one footprint-sized workgroup per CU).
inline int64_t pick_zchunk(int64_t nz, int64_t tiles, int blocks_per_cu) {
  const double slots = (double)cu_count() * blocks_per_cu;
  double best = -1.0;
  int64_t best_chunk = nz;
  for (int64_t nzc = 1; nzc <= nz; ++nzc) {
    const int64_t chunk = (nz + nzc - 1) / nzc;
    if (chunk < 8 && nzc > 1) break;
    if ((nz + chunk - 1) / chunk != nzc) continue;
    const double blocks = (double)tiles * nzc;
    const double rounds = (double)(int64_t)((blocks + slots - 1) / slots);
    const double fill = blocks / (rounds * slots);
    const double eff = fill * (double)chunk / (double)(chunk + 2);
    if (eff > best) { best = eff; best_chunk = chunk; }
  }
  return best_chunk;
}
Tuning g_tune2;

template <typename T>
inline bool al16(const T *a) {
  return (reinterpret_cast<uintptr_t>(a) & 15u) == 0;
}

template <typename T, int VEC, int WX, int WY, int RY, int WPE, bool HUBER, bool L1>
int launch2_f(const T *xbar_in, T *xbar_out, const T *x_in, T *x_out, const T *bt,
            const T *p_in, T *p_out, const Geom<T> &G, const PdScalars<T> &S1,
            const PdScalars<T> &S2, hipStream_t st) {
  constexpr int TXB = WX * 64 * VEC;
  constexpr int TYV = WY * RY - 2;
  const int64_t ntx = (G.nx <= TXB) ? 1 : (G.nx + (TXB - 2 * VEC) - 1) / (TXB - 2 * VEC);
  const int64_t nty = (G.ny + TYV - 1) / TYV;
  int64_t zchunk = g_tune2.zchunk;
  if (zchunk <= 0) zchunk = pick_zchunk(G.nz, ntx * nty, WX * WY >= 16 ? 1 : 2);
  if (zchunk > G.nz) zchunk = G.nz;
  const int64_t nzc = (G.nz + zchunk - 1) / zchunk;
  int64_t blocks = ntx * nty * nzc;
  int64_t slab = 0;
  if (g_tune2.xcd_map && nty >= 16) {
    slab = (nty + 7) / 8;
    blocks = 8 * slab * ntx * nzc;
  }
  if (blocks > 0x7fffffff) return NSOL_EINVAL;
  hipLaunchKernelGGL((k_pd_fused2<T, VEC, WX, WY, RY, WPE, HUBER, L1>),
                     dim3((unsigned)blocks),
                     dim3(WX * WY * 64), 0, st, xbar_in, xbar_out, x_in, x_out, bt,
                     p_in, p_out, G, S1, S2, (int)ntx, (int)nty, (int)zchunk,
                     (int)slab);
  return launch_status();
}

template <typename T, int VEC, int WX, int WY, int RY, int WPE>
int launch2(const T *xbar_in, T *xbar_out, const T *x_in, T *x_out, const T *bt,
            const T *p_in, T *p_out, const Geom<T> &G, const PdScalars<T> &S1,
            const PdScalars<T> &S2, hipStream_t st) {
#define NSOL_F(H, L)                                                            \
  launch2_f<T, VEC, WX, WY, RY, WPE, H, L>(xbar_in, xbar_out, x_in, x_out, bt,   \
                                           p_in, p_out, G, S1, S2, st)
  if (S1.huber) return S1.l1 ? NSOL_F(true, true) : NSOL_F(true, false);
  return S1.l1 ? NSOL_F(false, true) : NSOL_F(false, false);
#undef NSOL_F
}

// returns -2 if the two-iteration kernel does not apply to this problem
template <typename T>
int fused2_impl(const T *xbar_in, T *xbar_out, const T *x_in, T *x_out,
                const T *bt, const T *p_in, T *p_out, int ndim, int64_t nz,
                int64_t ny, int64_t nx, double wx, double wy, double wz,
                const double *sigma, const double *hden, const double *tau,
                const double *tl, const double *theta, int flags, void *stream) {
  NSOL_CHECK_GEOM(ndim, nz, ny, nx);
  if (!xbar_in || !xbar_out || !x_in || !x_out || !bt || !p_out ||
      xbar_in == xbar_out || p_in == p_out || x_in == x_out)
    return NSOL_EINVAL;
  constexpr int VW = 16 / sizeof(T);
  if (!g_tune2.enable || ndim != 3 || nx % VW != 0 || nx / VW < 64 || ny < 8 ||
      nz < 8 || !al16(xbar_in) || !al16(xbar_out) || !al16(x_in) ||
      !al16(x_out) || !al16(bt) || !al16(p_out) || (p_in && !al16(p_in)) ||
      (nz * ny * nx) % VW != 0)
    return -2;
  const Geom<T> G = make_geom<T>(ndim, nz, ny, nx, wx, wy, wz);
  PdScalars<T> S[2];
  for (int i = 0; i < 2; ++i) {
    S[i].sigma = (T)sigma[i]; S[i].hden = huber_den<T>(hden[i]); S[i].tau = (T)tau[i];
    S[i].tl = (T)tl[i]; S[i].one_plus_tl = prox_den<T>(tl[i]);
    S[i].theta = (T)theta[i];
    S[i].huber = (flags & NSOL_PD_REG_HUBER) ? 1 : 0;
    S[i].l1 = (flags & NSOL_PD_DATA_L1) ? 1 : 0;
    S[i].has_p = (i == 1 || p_in != nullptr) ? 1 : 0;
  }
  hipStream_t st = as_stream(stream);
#define NSOL_L2(WX, WY, RY, WPE)                                                \
  launch2<T, VW, WX, WY, RY, WPE>(xbar_in, xbar_out, x_in, x_out, bt, p_in, p_out,  \
                                  G, S[0], S[1], st)
  const bool wide = nx / VW > 64;
  switch (g_tune2.variant) {
    case 1: return wide ? NSOL_L2(2, 4, 2, 2) : NSOL_L2(1, 4, 2, 1);
    default: return wide ? NSOL_L2(2, 8, 1, 4) : NSOL_L2(1, 8, 1, 2);
  }
#undef NSOL_L2
}

}  // namespace nsol_pd2

extern "C" {

int nsol_hip_set_param_pd2(const char *name, int value) {
  if (!name) return NSOL_EINVAL;
  if (!strcmp(name, "pd2_zchunk")) nsol_pd2::g_tune2.zchunk = value;
  else if (!strcmp(name, "pd2_enable")) nsol_pd2::g_tune2.enable = value;
  else if (!strcmp(name, "pd2_variant")) nsol_pd2::g_tune2.variant = value;
  else if (!strcmp(name, "pd2_xcd_map")) nsol_pd2::g_tune2.xcd_map = value;
  else return NSOL_EINVAL;
  return 0;
}

int nsol_pd_fused2_iter_f32(const float *xbar_in, float *xbar_out,
                            const float *x_in, float *x_out, const float *bt,
                            const float *p_in, float *p_out, int ndim,
                            int64_t nz, int64_t ny, int64_t nx, double wx,
                            double wy, double wz, const double *sigma2,
                            const double *hden2, const double *tau2,
                            const double *tl2, const double *theta2, int flags,
                            void *stream) {
  return nsol_pd2::fused2_impl<float>(xbar_in, xbar_out, x_in, x_out, bt, p_in,
                                      p_out, ndim, nz, ny, nx, wx, wy, wz, sigma2,
                                      hden2, tau2, tl2, theta2, flags, stream);
}
int nsol_pd_fused2_iter_f64(const double *xbar_in, double *xbar_out,
                            const double *x_in, double *x_out, const double *bt,
                            const double *p_in, double *p_out, int ndim,
                            int64_t nz, int64_t ny, int64_t nx, double wx,
                            double wy, double wz, const double *sigma2,
                            const double *hden2, const double *tau2,
                            const double *tl2, const double *theta2, int flags,
                            void *stream) {
  return nsol_pd2::fused2_impl<double>(xbar_in, xbar_out, x_in, x_out, bt, p_in,
                                       p_out, ndim, nz, ny, nx, wx, wy, wz, sigma2,
                                       hden2, tau2, tl2, theta2, flags, stream);
}
}
