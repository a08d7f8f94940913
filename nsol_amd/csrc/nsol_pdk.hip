// K Chambolle-Pock iterations in ONE pass over memory (temporal blocking of
// depth K = 2 or 3) on 2-D tiled footprints.
//
// reference: K consecutive trips of the loop body primal_dual_solver.py:242-256.
//
// Per launch the kernel reads xbar, x, bt, p[3] once and writes p[3], x, xbar
// once: 11 words per voxel for K iterations.
//
// Footprints.  The NW*64 lanes of a workgroup form a flat `rows` x `lxb` grid
// (lxb lanes of VEC voxels per row; neither needs to divide the wave size), so
// the footprint can be made close to square: iteration k needs iteration-(k-1)
// values one voxel further out, hence a footprint loses K-1 voxels on every
// side and only the interior (rows - 2(K-1)) x (lxb*VEC - 2*HX) is stored.
// k_pd_fused2 uses full rows (512 x 8 at nx = 512: 25 % of the lanes recompute
// overlap already at K = 2); here nx = 512 is cut into 4 x 128 valid columns
// with 34-lane rows, 30 rows per workgroup.
//
// Pipeline along z (one barrier per plane).  At step s
//   stage 1     runs iteration n+1 on plane s exactly like k_pd_fused (old-data
//               halos from L1/L2), results stay in registers;
//   F_k, k>=2   finishes iteration n+k on plane s-(k-1): z-component of the
//               dual, K^T, prox, over-relaxation (it was waiting for
//               xbar^(k-1) of the plane above);
//   IP_k, k>=2  in-plane part of iteration n+k on plane s-(k-2): every lane
//               publishes xbar^(k-1), p_y^(k-1) and the last p_x^(k-1) of its
//               wave in LDS (double buffered), reads the four neighbours'
//               entries and forms p_x^(k), p_y^(k) and the in-plane K^T.
// The arithmetic per voxel is that of K launches of k_pd_fused in the same
// order, so results are bit-identical.
#include <string.h>

#include "nsol_common.hpp"
#include "nsol_pd_common.hpp"

using namespace nsol;

namespace nsol_pdk {

template <typename T, int K>
struct StageScalars {
  T sigma[K], hden[K], tau[K], tl[K], optl[K], theta[K];
  int has_p;
};

struct Tiling {
  int lxb;    // lanes per footprint row
  int rows;   // footprint rows
  int xv;     // valid voxels per tile along x; 0 = one tile spans the row
  int ntx, nty;
};

template <typename T, int VEC, int NW, int K, int WPE, bool HUBER, bool L1>
__global__ __launch_bounds__(NW * 64, WPE) void k_pd_fusedk(
    const T *__restrict__ xbar_in, T *__restrict__ xbar_out,
    const T *__restrict__ x_in, T *__restrict__ x_out,
    const T *__restrict__ bt, const T *__restrict__ p_in,
    T *__restrict__ p_out, Geom<T> G, StageScalars<T, K> S, Tiling Q, int zchunk,
    int slab) {
  constexpr int NT = NW * 64;
  constexpr int H = K - 1;
  constexpr int HX = ((H + VEC - 1) / VEC) * VEC;
  __shared__ __attribute__((aligned(16))) T s_xb[2][K - 1][NT * VEC];
  __shared__ __attribute__((aligned(16))) T s_py[2][K - 1][NT * VEC];
  __shared__ T s_px[2][K - 1][NW];   // last p_x^(k-1) of each wave's lane 63

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // Block -> (tile, z-chunk).  blockIdx % 8 names the XCD under the round-robin
  // dispatch; each XCD walks `slab` consecutive tiles so that footprints which
  // overlap mostly meet in one L2.  A whole workgroup leaves together.
  const int ntiles = Q.ntx * Q.nty;
  int tile, zc;
  if (slab > 0) {
    const int xcd = blockIdx.x & 7;
    const int j = blockIdx.x >> 3;
    tile = xcd * slab + j % slab;
    zc = j / slab;
    if (tile >= ntiles) return;
  } else {
    tile = blockIdx.x % ntiles;
    zc = blockIdx.x / ntiles;
  }
  const int tx = tile % Q.ntx;
  const int ty = tile / Q.ntx;

  // ---- geometry of this lane
  const int lxb = Q.lxb;
  const int row = tid / lxb;
  const int lx = tid - row * lxb;
  const bool active = row < Q.rows;
  const bool single_x = (Q.xv == 0);
  const int64_t xv_lo = single_x ? 0 : (int64_t)tx * Q.xv;
  const int64_t xv_hi = single_x ? G.nx : xv_lo + Q.xv;
  const int64_t x0 = (single_x ? 0 : xv_lo - HX) + (int64_t)lx * VEC;
  const int64_t tyv = Q.rows - 2 * H;
  const int64_t yv_lo = (int64_t)ty * tyv;
  const int64_t yv_hi = yv_lo + tyv;
  const int64_t y = yv_lo - H + row;
  const bool rin = active && x0 >= 0 && x0 < G.nx && y >= 0 && y < G.ny;
  const bool rvalid = rin && x0 >= xv_lo && x0 < xv_hi && y >= yv_lo && y < yv_hi;

  // Step sizes masked to zero outside the volume: a voxel there gets p = 0 and
  // x = xbar = 0 in every stage (K and K^T pad with zeros) without selects.
  T sig_m[K], tau_m[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    sig_m[k] = rin ? S.sigma[k] : T(0);
    tau_m[k] = rin ? S.tau[k] : T(0);
  }

  const int64_t zbeg = (int64_t)zc * zchunk;
  int64_t zend = zbeg + zchunk;
  if (zend > G.nz) zend = G.nz;

  const T *pin_x = p_in, *pin_y = p_in + G.n, *pin_z = p_in + 2 * G.n;
  T *pout_x = p_out, *pout_y = p_out + G.n, *pout_z = p_out + 2 * G.n;

  // stage 1 halos (old data, global memory)
  const bool row_end = (lx == lxb - 1) || lane == 63;
  const bool row_beg = (lx == 0) || lane == 0;
  const bool g_right = rin && row_end && (x0 + VEC < G.nx);
  const bool g_left = rin && row_beg && x0 > 0;
  const bool g_up = rin && y > 0;
  const bool g_down = rin && (y + 1 < G.ny);
  // stages >= 2: neighbours inside the footprint (LDS); beyond it: zero (either
  // the zero padding of K / K^T at the volume edge, or a voxel whose result is
  // recomputed by the neighbouring footprint and never stored from here)
  const bool n_l = active && lx > 0;
  const bool n_r = active && lx < lxb - 1;
  const bool n_u = active && row > 0;
  const bool n_d = active && row + 1 < Q.rows;
  const bool v_l = rin && n_l && x0 > 0;   // the left / upper voxel is in the volume
  const bool v_u = rin && n_u && y > 0;

  const int64_t s_first = zbeg > H ? zbeg - H : 0;
  const int64_t s_last = zend + K - 2;
  int64_t off = s_first * G.sz + y * G.sy + x0;

  T xc[VEC];              // xbar[s]
  T pz[K][VEC];           // pz[k-1]: p^(k)_z on the plane stage k finished last
  T c_xb[K][VEC];         // [k-1], k>=2: xbar^(k-1) on plane s-(k-1)
  T c_x[K][VEC];          //               x^(k-1) there
  T c_bt[K][VEC];         //               bt there
  T c_kt[K][VEC];         //               in-plane part of K^T p^(k) there
  T c_px[K][VEC];         // [k-1], k>=3: p_x^(k-1) on plane s-(k-1)+1 (from IP_{k-1})
  T c_py[K][VEC];
  zero(xc);
#pragma unroll
  for (int k = 0; k < K; ++k) {
    zero(pz[k]); zero(c_xb[k]); zero(c_x[k]); zero(c_bt[k]); zero(c_kt[k]);
    zero(c_px[k]); zero(c_py[k]);
  }
  if (rin) ldv<T, VEC>(xbar_in + off, xc);
  if (s_first > 0 && rin) {
    T xm[VEC], pm[VEC];
    zero(pm);
    ldv<T, VEC>(xbar_in + off - G.sz, xm);
    if (S.has_p) ldv<T, VEC>(pin_z + off - G.sz, pm);
#pragma unroll
    for (int j = 0; j < VEC; ++j)
      pz[0][j] = dual_update_s<HUBER>(pm[j], xc[j], xm[j], G.wz, S.sigma[0], S.hden[0]);
  }

  // Software pipeline: the global loads of plane s+1 are issued right after the
  // stage-1 arithmetic of plane s, so they are in flight while the later stages
  // (and the barrier) run -- the waves of a workgroup move in lockstep, so
  // nothing else would hide the latency.
  T xn[VEC], xv[VEC], btn[VEC], pxo[VEC], pyo[VEC], pzo[VEC];
  T xdown[VEC], xup[VEC], pyup[VEC];
  T xright = T(0), xleft = T(0), pxleft = T(0);
  zero(xn); zero(xv); zero(btn); zero(pxo); zero(pyo); zero(pzo);
  zero(xdown); zero(xup); zero(pyup);
  auto issue_loads = [&](int64_t sp, int64_t o) {
    if (rin) {
      if (sp + 1 < G.nz) ldv<T, VEC>(xbar_in + o + G.sz, xn);
      else zero(xn);
      ldv<T, VEC>(x_in + o, xv);
      ldv<T, VEC>(bt + o, btn);
      if (S.has_p) {
        ldv<T, VEC>(pin_x + o, pxo);
        ldv<T, VEC>(pin_y + o, pyo);
        ldv<T, VEC>(pin_z + o, pzo);
      }
    }
    if (g_right) xright = xbar_in[o + VEC];
    if (g_left) {
      xleft = xbar_in[o - 1];
      if (S.has_p) pxleft = pin_x[o - 1];
    }
    if (g_down) ldv<T, VEC>(xbar_in + o + G.sy, xdown);
    if (g_up) {
      ldv<T, VEC>(xbar_in + o - G.sy, xup);
      if (S.has_p) ldv<T, VEC>(pin_y + o - G.sy, pyup);
    }
  };
  if (s_first < G.nz) issue_loads(s_first, off);

  for (int64_t s = s_first; s <= s_last; ++s, off += G.sz) {
    // fr_*[k-1]: results of stage k produced in this step
    T fr_xb[K][VEC], fr_x[K][VEC], fr_bt[K][VEC], pzn[K][VEC];
    T f_px[VEC], f_py[VEC];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      zero(fr_xb[k]); zero(fr_x[k]); zero(fr_bt[k]); zero(pzn[k]);
    }
    zero(f_px); zero(f_py);

    if (s < G.nz) {
      // ================= stage 1: iteration n+1 on plane s ===================
#pragma unroll
      for (int j = 0; j < VEC; ++j) fr_bt[0][j] = btn[j];
      T nb = __shfl_down(xc[0], 1, kWave);
      if (row_end) nb = xright;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const T hx = (j + 1 < VEC) ? xc[(j + 1) % VEC] : nb;
        f_px[j] = dual_update_s<HUBER>(pxo[j], hx, xc[j], G.wx, sig_m[0], S.hden[0]);
        f_py[j] = dual_update_s<HUBER>(pyo[j], xdown[j], xc[j], G.wy, sig_m[0], S.hden[0]);
        pzn[0][j] = dual_update_s<HUBER>(pzo[j], xn[j], xc[j], G.wz, sig_m[0], S.hden[0]);
      }
      T pxl = __shfl_up(f_px[VEC - 1], 1, kWave);
      if (row_beg)
        pxl = g_left ? dual_update_s<HUBER>(pxleft, xc[0], xleft, G.wx, S.sigma[0],
                                            S.hden[0])
                     : T(0);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const T pu = g_up ? dual_update_s<HUBER>(pyup[j], xc[j], xup[j], G.wy,
                                                 S.sigma[0], S.hden[0])
                          : T(0);
        const T pl = (j > 0) ? f_px[(j + VEC - 1) % VEC] : pxl;
        T kt = f_px[j] * (-G.wx) + pl * G.wx;
        kt += f_py[j] * (-G.wy) + pu * G.wy;
        kt += pzn[0][j] * (-G.wz) + pz[0][j] * G.wz;
        const T u = xv[j] - tau_m[0] * kt;
        const T xnew = prox_data_s<L1>(u, fr_bt[0][j], S.tl[0], S.optl[0]);
        fr_x[0][j] = xnew;
        fr_xb[0][j] = xnew + S.theta[0] * (xnew - xv[j]);
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) xc[j] = xn[j];
      if (s + 1 < G.nz && s + 1 <= s_last) issue_loads(s + 1, off + G.sz);
    }

    // ================= F_k: finish iteration n+k on plane s-(k-1) ===========
#pragma unroll
    for (int k = 2; k <= K; ++k) {
      const int64_t f = s - (k - 1);
      if (f >= s_first) {
        // beyond the last plane everything is zero padding (only k < K gets there)
        const T tk = (f < G.nz) ? tau_m[k - 1] : T(0);
        T pkz[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          pkz[j] = dual_update_s<HUBER>(pz[k - 2][j], fr_xb[k - 2][j], c_xb[k - 1][j],
                                        G.wz, sig_m[k - 1], S.hden[k - 1]);
          T kt = c_kt[k - 1][j];
          kt += pkz[j] * (-G.wz) + pz[k - 1][j] * G.wz;
          const T u = c_x[k - 1][j] - tk * kt;
          const T xk = prox_data_s<L1>(u, c_bt[k - 1][j], S.tl[k - 1], S.optl[k - 1]);
          fr_x[k - 1][j] = xk;
          fr_xb[k - 1][j] = xk + S.theta[k - 1] * (xk - c_x[k - 1][j]);
          fr_bt[k - 1][j] = c_bt[k - 1][j];
          pzn[k - 1][j] = pkz[j];
        }
        if (k == K && f >= zbeg && f < zend && rvalid) {
          const int64_t o = off - (int64_t)(K - 1) * G.sz;
          stv<T, VEC>(pout_z + o, pkz);
          stv<T, VEC>(x_out + o, fr_x[K - 1]);
          stv<T, VEC>(xbar_out + o, fr_xb[K - 1]);
        }
      }
    }

    // ================= IP_k: in-plane part of iteration n+k on plane s-(k-2) =
    const int buf = (int)(s & 1);
#pragma unroll
    for (int k = 2; k <= K; ++k) {
      stv<T, VEC>(&s_xb[buf][k - 2][tid * VEC], fr_xb[k - 2]);
      if (k == 2) {
        stv<T, VEC>(&s_py[buf][0][tid * VEC], f_py);
        if (lane == 63) s_px[buf][0][tid >> 6] = f_px[VEC - 1];
      } else {
        stv<T, VEC>(&s_py[buf][k - 2][tid * VEC], c_py[k - 1]);
        if (lane == 63) s_px[buf][k - 2][tid >> 6] = c_px[k - 1][VEC - 1];
      }
    }
    __syncthreads();
    T n_kt[K][VEC], n_px[K][VEC], n_py[K][VEC];
#pragma unroll
    for (int k = 2; k <= K; ++k) {
      T below[VEC], above[VEC], above_py[VEC];
      zero(below); zero(above); zero(above_py);
      T right = T(0), left_xb = T(0), left_px = T(0);
      if (n_d) ldv<T, VEC>(&s_xb[buf][k - 2][(tid + lxb) * VEC], below);
      if (n_u) {
        ldv<T, VEC>(&s_xb[buf][k - 2][(tid - lxb) * VEC], above);
        ldv<T, VEC>(&s_py[buf][k - 2][(tid - lxb) * VEC], above_py);
      }
      if (n_r) right = s_xb[buf][k - 2][(tid + 1) * VEC];
      left_px = __shfl_up((k == 2) ? f_px[VEC - 1] : c_px[k - 1][VEC - 1], 1, kWave);
      if (n_l) {
        left_xb = s_xb[buf][k - 2][tid * VEC - 1];
        if (lane == 0) left_px = s_px[buf][k - 2][(tid >> 6) - 1];
      }
      const T pl0 = v_l ? dual_update_s<HUBER>(left_px, fr_xb[k - 2][0], left_xb, G.wx,
                                               S.sigma[k - 1], S.hden[k - 1])
                        : T(0);
      T pkx[VEC], pky[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const T px_old = (k == 2) ? f_px[j] : c_px[k - 1][j];
        const T py_old = (k == 2) ? f_py[j] : c_py[k - 1][j];
        const T hx = (j + 1 < VEC) ? fr_xb[k - 2][(j + 1) % VEC] : right;
        pkx[j] = dual_update_s<HUBER>(px_old, hx, fr_xb[k - 2][j], G.wx, sig_m[k - 1],
                                      S.hden[k - 1]);
        pky[j] = dual_update_s<HUBER>(py_old, below[j], fr_xb[k - 2][j], G.wy,
                                      sig_m[k - 1], S.hden[k - 1]);
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const T pu = v_u ? dual_update_s<HUBER>(above_py[j], fr_xb[k - 2][j], above[j],
                                                G.wy, S.sigma[k - 1], S.hden[k - 1])
                         : T(0);
        const T pl = (j > 0) ? pkx[(j + VEC - 1) % VEC] : pl0;
        T kt = pkx[j] * (-G.wx) + pl * G.wx;
        kt += pky[j] * (-G.wy) + pu * G.wy;
        n_kt[k - 1][j] = kt;
        n_px[k - 1][j] = pkx[j];
        n_py[k - 1][j] = pky[j];
      }
      if (k == K) {
        const int64_t a = s - (K - 2);
        if (a >= zbeg && a < zend && rvalid) {
          const int64_t o = off - (int64_t)(K - 2) * G.sz;
          stv<T, VEC>(pout_x + o, pkx);
          stv<T, VEC>(pout_y + o, pky);
        }
      }
    }

    // ================= carry ===============================================
#pragma unroll
    for (int k = K; k >= 2; --k) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        c_xb[k - 1][j] = fr_xb[k - 2][j];
        c_x[k - 1][j] = fr_x[k - 2][j];
        c_bt[k - 1][j] = fr_bt[k - 2][j];
        c_kt[k - 1][j] = n_kt[k - 1][j];
        if (k < K) {
          c_px[k][j] = n_px[k - 1][j];
          c_py[k][j] = n_py[k - 1][j];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int j = 0; j < VEC; ++j) pz[k][j] = pzn[k][j];
  }
}

struct Tuning {
  int enable = 0;
  int kmax = 3;
  int nw = 16;
  int zchunk = 0;
  int ntx = 0;      // 0 = choose
  int xcd_map = 1;
};
Tuning g_tunek;

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

// Footprint shape with the largest share of lanes that produce stored output.
inline bool choose_tiling(int64_t nx, int64_t ny, int nt, int vec, int h, int hx,
                          int force_ntx, Tiling *out) {
  double best = 0.0;
  bool found = false;
  for (int ntx = 1; ntx <= 64; ++ntx) {
    if (force_ntx > 0 && ntx != force_ntx) continue;
    Tiling q;
    if (ntx == 1) {
      q.xv = 0;
      q.lxb = (int)(nx / vec);
    } else {
      int64_t xv = (nx + ntx - 1) / ntx;
      xv = (xv + vec - 1) / vec * vec;
      if (xv * (ntx - 1) >= nx) continue;   // fewer tiles would do
      q.xv = (int)xv;
      q.lxb = (int)(xv / vec) + 2 * (hx / vec);
    }
    if (q.lxb < 1 || q.lxb > nt) continue;
    q.rows = nt / q.lxb;
    if (q.rows - 2 * h < 1) continue;
    q.ntx = ntx;
    q.nty = (int)((ny + (q.rows - 2 * h) - 1) / (q.rows - 2 * h));
    const double eff = (double)nx * (double)ny /
                       ((double)q.ntx * q.nty * (double)nt * vec);
    if (eff > best) { best = eff; *out = q; found = true; }
  }
  return found;
}

// z-chunk length: trade the 2(K-1) extra planes per chunk against filling the
// last round of workgroups (one workgroup per CU).
inline int64_t pick_zchunk(int64_t nz, int64_t tiles, int extra) {
  const double slots = (double)cu_count();
  double best = -1.0;
  int64_t best_chunk = nz;
  for (int64_t nzc = 1; nzc <= nz; ++nzc) {
    const int64_t chunk = (nz + nzc - 1) / nzc;
    if (chunk < 8 && nzc > 1) break;
    if ((nz + chunk - 1) / chunk != nzc) continue;
    const double blocks = (double)tiles * nzc;
    const double rounds = (double)(int64_t)((blocks + slots - 1) / slots);
    const double fill = blocks / (rounds * slots);
    const double eff = fill * (double)chunk / (double)(chunk + extra);
    if (eff > best) { best = eff; best_chunk = chunk; }
  }
  return best_chunk;
}

template <typename T>
inline bool al16(const T *a) {
  return (reinterpret_cast<uintptr_t>(a) & 15u) == 0;
}

template <typename T, int VEC, int NW, int K, int WPE, bool HUBER, bool L1>
int launch_f(const T *xbar_in, T *xbar_out, const T *x_in, T *x_out, const T *bt,
             const T *p_in, T *p_out, const Geom<T> &G, const StageScalars<T, K> &S,
             hipStream_t st) {
  constexpr int H = K - 1;
  constexpr int HX = ((H + VEC - 1) / VEC) * VEC;
  Tiling Q;
  if (!choose_tiling(G.nx, G.ny, NW * 64, VEC, H, HX, g_tunek.ntx, &Q)) return -2;
  const int64_t tiles = (int64_t)Q.ntx * Q.nty;
  int64_t zchunk = g_tunek.zchunk;
  if (zchunk <= 0) zchunk = pick_zchunk(G.nz, tiles, 2 * H);
  if (zchunk > G.nz) zchunk = G.nz;
  const int64_t nzc = (G.nz + zchunk - 1) / zchunk;
  int64_t blocks = tiles * nzc;
  int64_t slab = 0;
  if (g_tunek.xcd_map && tiles >= 16) {
    slab = (tiles + 7) / 8;
    blocks = 8 * slab * nzc;
  }
  if (blocks > 0x7fffffff) return NSOL_EINVAL;
  hipLaunchKernelGGL((k_pd_fusedk<T, VEC, NW, K, WPE, HUBER, L1>),
                     dim3((unsigned)blocks), dim3(NW * 64), 0, st, xbar_in, xbar_out,
                     x_in, x_out, bt, p_in, p_out, G, S, Q, (int)zchunk, (int)slab);
  return launch_status();
}

template <typename T, int VEC, int NW, int K, int WPE>
int launch_k(const T *xbar_in, T *xbar_out, const T *x_in, T *x_out, const T *bt,
             const T *p_in, T *p_out, const Geom<T> &G, const StageScalars<T, K> &S,
             int flags, hipStream_t st) {
#define NSOL_F(HB, L)                                                           \
  launch_f<T, VEC, NW, K, WPE, HB, L>(xbar_in, xbar_out, x_in, x_out, bt, p_in,  \
                                      p_out, G, S, st)
  const bool huber = (flags & NSOL_PD_REG_HUBER) != 0;
  const bool l1 = (flags & NSOL_PD_DATA_L1) != 0;
  if (huber) return l1 ? NSOL_F(true, true) : NSOL_F(true, false);
  return l1 ? NSOL_F(false, true) : NSOL_F(false, false);
#undef NSOL_F
}

template <typename T, int K>
int fusedk_k(const T *xbar_in, T *xbar_out, const T *x_in, T *x_out, const T *bt,
             const T *p_in, T *p_out, const Geom<T> &G, const double *sigma,
             const double *hden, const double *tau, const double *tl,
             const double *theta, int flags, hipStream_t st) {
  constexpr int VW = 16 / sizeof(T);
  StageScalars<T, K> S;
  for (int i = 0; i < K; ++i) {
    S.sigma[i] = (T)sigma[i]; S.hden[i] = (T)hden[i]; S.tau[i] = (T)tau[i];
    S.tl[i] = (T)tl[i]; S.optl[i] = prox_den<T>(tl[i]); S.theta[i] = (T)theta[i];
  }
  S.has_p = p_in != nullptr ? 1 : 0;
  if (g_tunek.nw == 8)
    return launch_k<T, VW, 8, K, 2>(xbar_in, xbar_out, x_in, x_out, bt, p_in, p_out,
                                    G, S, flags, st);
  if (g_tunek.nw == 12)
    return launch_k<T, VW, 12, K, 3>(xbar_in, xbar_out, x_in, x_out, bt, p_in, p_out,
                                     G, S, flags, st);
  return launch_k<T, VW, 16, K, 4>(xbar_in, xbar_out, x_in, x_out, bt, p_in, p_out, G,
                                   S, flags, st);
}

// returns -2 if the kernel does not apply to this problem
template <typename T>
int fusedk_impl(const T *xbar_in, T *xbar_out, const T *x_in, T *x_out, const T *bt,
                const T *p_in, T *p_out, int ndim, int64_t nz, int64_t ny,
                int64_t nx, double wx, double wy, double wz, int k,
                const double *sigma, const double *hden, const double *tau,
                const double *tl, const double *theta, int flags, void *stream) {
  NSOL_CHECK_GEOM(ndim, nz, ny, nx);
  if (!xbar_in || !xbar_out || !x_in || !x_out || !bt || !p_out || !sigma ||
      !hden || !tau || !tl || !theta || xbar_in == xbar_out || p_in == p_out ||
      x_in == x_out || k < 2 || k > 3)
    return NSOL_EINVAL;
  constexpr int VW = 16 / sizeof(T);
  if (!g_tunek.enable || k > g_tunek.kmax || ndim != 3 || nx % VW != 0 ||
      nx / VW < 8 || ny < 8 || nz < 8 || !al16(xbar_in) || !al16(xbar_out) ||
      !al16(x_in) || !al16(x_out) || !al16(bt) || !al16(p_out) ||
      (p_in && !al16(p_in)) || (nz * ny * nx) % VW != 0)
    return -2;
  const Geom<T> G = make_geom<T>(ndim, nz, ny, nx, wx, wy, wz);
  hipStream_t st = as_stream(stream);
  if (k == 2)
    return fusedk_k<T, 2>(xbar_in, xbar_out, x_in, x_out, bt, p_in, p_out, G, sigma,
                          hden, tau, tl, theta, flags, st);
  return fusedk_k<T, 3>(xbar_in, xbar_out, x_in, x_out, bt, p_in, p_out, G, sigma,
                        hden, tau, tl, theta, flags, st);
}

}  // namespace nsol_pdk

extern "C" {

int nsol_hip_set_param_pdk(const char *name, int value) {
  if (!name) return NSOL_EINVAL;
  if (!strcmp(name, "pdk_enable")) nsol_pdk::g_tunek.enable = value;
  else if (!strcmp(name, "pdk_kmax")) nsol_pdk::g_tunek.kmax = value;
  else if (!strcmp(name, "pdk_nw")) nsol_pdk::g_tunek.nw = value;
  else if (!strcmp(name, "pdk_zchunk")) nsol_pdk::g_tunek.zchunk = value;
  else if (!strcmp(name, "pdk_ntx")) nsol_pdk::g_tunek.ntx = value;
  else if (!strcmp(name, "pdk_xcd_map")) nsol_pdk::g_tunek.xcd_map = value;
  else return NSOL_EINVAL;
  return 0;
}

int nsol_pd_fusedk_iter_f32(const float *xbar_in, float *xbar_out, const float *x_in,
                            float *x_out, const float *bt, const float *p_in,
                            float *p_out, int ndim, int64_t nz, int64_t ny,
                            int64_t nx, double wx, double wy, double wz, int k,
                            const double *sigma, const double *hden,
                            const double *tau, const double *tl,
                            const double *theta, int flags, void *stream) {
  return nsol_pdk::fusedk_impl<float>(xbar_in, xbar_out, x_in, x_out, bt, p_in, p_out,
                                      ndim, nz, ny, nx, wx, wy, wz, k, sigma, hden,
                                      tau, tl, theta, flags, stream);
}
int nsol_pd_fusedk_iter_f64(const double *xbar_in, double *xbar_out,
                            const double *x_in, double *x_out, const double *bt,
                            const double *p_in, double *p_out, int ndim, int64_t nz,
                            int64_t ny, int64_t nx, double wx, double wy, double wz,
                            int k, const double *sigma, const double *hden,
                            const double *tau, const double *tl,
                            const double *theta, int flags, void *stream) {
  return nsol_pdk::fusedk_impl<double>(xbar_in, xbar_out, x_in, x_out, bt, p_in,
                                       p_out, ndim, nz, ny, nx, wx, wy, wz, k, sigma,
                                       hden, tau, tl, theta, flags, stream);
}
}
