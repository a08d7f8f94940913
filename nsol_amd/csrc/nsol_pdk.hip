// K Chambolle-Pock iterations in ONE pass over memory (temporal blocking of
// depth K = 2 or 3) on 2-D tiled footprints.
//
// reference: K consecutive trips of the loop body primal_dual_solver.py:242-256.
//
// Per launch the kernel reads xbar, x, bt, p[3] once and writes p[3], x, xbar
// once: 11 words per voxel for K iterations.
//
// Footprints.  The NW*64 lanes of a workgroup form a `rows` x `lxb` grid (lxb lanes
// of VEC voxels per row), so the footprint can be made close to square:
// iteration k needs iteration-(k-1) values one voxel further out, hence a
// footprint loses K-1 voxels on every side and only the interior
// (rows - 2(K-1)) x (lxb*VEC - 2*HX) is stored; the overlap is recomputed, never
// exchanged between workgroups.  k_pd_fused2 uses full rows (512 x 8 at
// nx = 512: 25 % of the lanes recompute overlap already at K = 2); here nx = 512
// is cut into 2 x 256 valid columns: 64 interior + 2 halo lanes per row, 11 rows
// per 12-wave workgroup, the interior lanes of a row in one wave ("split" lane
// mapping) and the halo lanes of all rows together in the last wave.
//
// Pipeline along z (one barrier per plane).  At step s
//   stage 1     runs iteration n+1 on plane s exactly like k_pd_fused (old-data
//               halos from L1/L2), results stay in registers; the loads of plane
//               s+1 are issued right behind it (software prefetch);
//   F_k, k>=2   finishes iteration n+k on plane s-(k-1): z-component of the
//               dual, K^T, prox, over-relaxation (it was waiting for
//               xbar^(k-1) of the plane above);
//   IP_k, k>=2  in-plane part of iteration n+k on plane s-(k-2): every lane
//               publishes xbar^(k-1), p_y^(k-1) and the last p_x^(k-1) of its
//               vector in LDS (double buffered, addressed by logical position),
//               reads the four neighbours' entries and forms p_x^(k), p_y^(k)
//               and the in-plane K^T.
// Waves whose rows lie so far out that nobody uses their stage-k results skip
// that arithmetic.  All global accesses are raw buffer loads / stores with a
// loop-invariant per-lane offset (out of range for lanes outside the volume), so
// the loop body has no exec-mask branches.  The arithmetic per voxel is that of K
// launches of k_pd_fused in the same order, so results are bit-identical.
//
// Host side (bottom of the file): a cost model ranks footprint shapes; for large
// volumes an online tuner measures the best few on the run's own launches.
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <deque>
#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>
#include <vector>

#include "nsol_common.hpp"
#include "nsol_pd_common.hpp"

using namespace nsol;

// Timing experiments only (wrong results): -DPDK_ABLATE=32 no stage-1 halo loads, =1 no global loads,
// 2 no global stores, 4 no barrier, 8 no arithmetic (memory pattern + LDS exchange
// + barrier only), 16 with 8: no LDS exchange and no barrier either; bits combine
// (DESIGN.md section 5).
// Stage 1's halo (the old xbar of the rows above / below, the old p_y of the row above,
// the x neighbours of a wave's first / last lane) exchanged through the LDS inside the
// 12-wave depth-3 workgroup instead of re-loaded from global memory; only the footprint's
// outermost rows and lanes still load theirs (0: every lane loads its halo, for A/B runs)
#ifndef PDK_LH
#define PDK_LH 1
#endif
#ifndef PDK_ABLATE
#define PDK_ABLATE 0
#endif
// wave priority while a step's loads (bit 0) / stores (bit 1) are issued (experiment)
#ifndef PDK_PRIO
#define PDK_PRIO 0
#endif
// cache policy of the 16-byte plane loads (experiments: 1 = sc0, 2 = nt, 16 = sc1)
#ifndef PDK_LOAD_AUX
#define PDK_LOAD_AUX 0
#endif

namespace nsol_pdk {

template <typename T, int K>
struct StageScalars {
  T sigma[K], hden[K], tau[K], tl[K], optl[K], theta[K];
  int has_p;
};

// launches of k_pd_fusedk so far, [0]: K = 2, [1]: K = 3 (tests ask which kernel ran)
std::atomic<int> g_launches[2];

struct Tiling {
  int lxb;    // lanes per footprint row
  int rows;   // footprint rows
  int xv;     // valid voxels per tile along x; 0 = one tile spans the row
  int ntx, nty;
  int split;  // 1: whole waves hold the interior lanes of a row, halo lanes at the end
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

// Raw buffer addressing: a 32-bit byte offset per lane plus a scalar offset,
// and the hardware range check drops accesses whose offset is >= num_records
// (loads return 0).  Lanes outside the volume carry kInvalid as their offset,
// so the loop body needs neither exec-mask branches nor zero-initialised
// destination registers -- which is what lets the prefetch stay in flight.
constexpr uint32_t kInvalid = 0xC0000000u;

__device__ __forceinline__ rsrc_t make_rsrc(const void *p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
template <typename T, int V>
__device__ __forceinline__ void bld(rsrc_t r, uint32_t vo, uint32_t so, T (&v)[V]) {
  static_assert(sizeof(T) * V == 16, "one 16-byte vector per lane");
  typedef typename Pack<T, V>::type P;
#if PDK_ABLATE & 1
  for (int k = 0; k < V; ++k) v[k] = T(1e-3) * (T)((vo + so + k) & 255u);
  return;
#endif
  const P t = __builtin_bit_cast(P, __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, PDK_LOAD_AUX));
#pragma unroll
  for (int k = 0; k < V; ++k) v[k] = t[k];
}
template <typename T> __device__ __forceinline__ T bld1(rsrc_t r, uint32_t vo, uint32_t so);
template <> __device__ __forceinline__ float bld1<float>(rsrc_t r, uint32_t vo, uint32_t so) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, vo, so, 0));
}
template <> __device__ __forceinline__ double bld1<double>(rsrc_t r, uint32_t vo, uint32_t so) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, 0));
}
// The whole offset goes into the VGPR and the scalar offset is the constant 0.
// A 16-byte buffer store reads its data registers over two passes; when the
// scalar offset is a REGISTER the compiler (ROCm 7.2) assumes there is no hazard
// and lets the next VALU instruction overwrite them -- on gfx950 that corrupted
// the stored vector in the lanes of the later passes.  With a constant scalar
// offset its hazard recogniser keeps a wait state between the store and such a
// write (checked in the ISA: at least one instruction separates them); the
// s_nop below is belt and braces only -- the scheduler is free to move it away
// from the store.
// The five output streams are written once and not read again by this launch:
// non-temporal stores (aux bit 1) leave the L2 to the footprints' overlap rows, which
// neighbouring workgroups do re-read (+1.1 % on one box, A/B with tools/_probe/ab_lib.sh).
constexpr int kStoreAux = 2;
template <typename T, int V>
__device__ __forceinline__ void bst(rsrc_t r, uint32_t vo, const T (&v)[V]) {
  typedef typename Pack<T, V>::type P;
  P t;
#pragma unroll
  for (int k = 0; k < V; ++k) t[k] = v[k];
#if PDK_ABLATE & 2
  if (t[0] != T(123.456)) return;
#endif
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, t), r, vo, 0, kStoreAux);
  asm volatile("s_nop 1");
}

// the first n elements only (the last vector of a row that is not a multiple of
// the vector width); same rules for the offsets as bst()
template <typename T, int V>
__device__ __forceinline__ void bst_part(rsrc_t r, uint32_t vo, const T (&v)[V], int n) {
#pragma unroll
  for (int j = 0; j < V; ++j) {
    if (j < n) {
      if constexpr (sizeof(T) == 4)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v[j]), r,
                                              vo + (uint32_t)(j * sizeof(T)), 0, 0);
      else
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v[j]), r,
                                              vo + (uint32_t)(j * sizeof(T)), 0, 0);
    }
  }
  asm volatile("s_nop 1");
}

// Element-wise dual update of a whole vector, out[j] = clamp((p_old[j] + sigma *
// (hi[j] - lo[j]) * w) / hden), with the float arithmetic issued as packed pairs
// (v_pk_add_f32 / v_pk_mul_f32).  There is no packed subtract and the compiler
// turns a + (-b) back into a scalar v_sub, hence the one-line asm with the
// negation in the operand modifiers; all values are exactly those of
// dual_update_u.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <bool HUBER, bool UNIT, typename T, int V>
__device__ __forceinline__ void dual_vec(T (&out)[V], const T (&p_old)[V],
                                         const T (&hi)[V], const T (&lo)[V], T w,
                                         T sigma, T hden) {
  if constexpr (sizeof(T) == 4 && V % 2 == 0) {
#pragma unroll
    for (int h = 0; h < V / 2; ++h) {
      const f32x2 a = {hi[2 * h], hi[2 * h + 1]};
      const f32x2 b = {lo[2 * h], lo[2 * h + 1]};
      const f32x2 p = {p_old[2 * h], p_old[2 * h + 1]};
      const f32x2 g = UNIT ? pk_sub(a, b) : a * w + b * (-w);
      f32x2 q = p + sigma * g;
      if constexpr (HUBER) q = q * hden;      // float: hden holds the reciprocal
      out[2 * h] = dual_clamp(q[0]);
      out[2 * h + 1] = dual_clamp(q[1]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < V; ++j)
      out[j] = dual_update_u<HUBER, UNIT>(p_old[j], hi[j], lo[j], w, sigma, hden);
  }
}

// LDS rows are padded by one (zero) row above and below the footprint
// lane in `mask` ? s : 0, formed where it is used: volatile, so that the compiler neither
// hoists it out of the plane loop nor keeps it in a register across a step (the
// LDS-halo form of the kernel has no register to spare for nine such per-lane constants)
__device__ __forceinline__ float masked_scalar(float s, uint64_t mask) {
  float r;
  // (one scalar operand per VOP3 instruction on gfx9: the value moves first)
  asm volatile("v_mov_b32 %0, %1\n\tv_cndmask_b32_e64 %0, 0, %0, %2"
               : "=&v"(r) : "s"(s), "s"(mask));
  return r;
}
__device__ __forceinline__ double masked_scalar(double s, uint64_t mask) {
  const uint64_t b = __builtin_bit_cast(uint64_t, s);
  uint32_t lo, hi;
  asm volatile("v_mov_b32 %0, %1\n\tv_cndmask_b32_e64 %0, 0, %0, %2"
               : "=&v"(lo) : "s"((uint32_t)b), "s"(mask));
  asm volatile("v_mov_b32 %0, %1\n\tv_cndmask_b32_e64 %0, 0, %0, %2"
               : "=&v"(hi) : "s"((uint32_t)(b >> 32)), "s"(mask));
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

template <int NT, int K>
struct LdsShape {
  static constexpr int LXM = NT / (2 * K);          // widest row (lanes)
  static constexpr int N = NT + 2 * LXM;            // vectors per array
};

template <typename T, int V>
struct PlaneLoads {          // registers one prefetched plane lands in
  T xn[V], xv[V], btn[V], pxo[V], pyo[V], pzo[V], xdown[V], xup[V], pyup[V];
  T xright, xleft, pxleft;
};

template <typename T, int VEC, int NW, int K, int WPE, bool HUBER, bool L1, bool PF2,
          bool UNIT, bool RAG>
__global__ __launch_bounds__(NW * 64, WPE) void k_pd_fusedk(
    const T *__restrict__ xbar_in, T *__restrict__ xbar_out,
    const T *__restrict__ x_in, T *__restrict__ x_out,
    const T *__restrict__ bt, const T *__restrict__ p_in,
    T *__restrict__ p_out, Geom<T> G, StageScalars<T, K> S, Tiling Q, int zchunk,
    int slab) {
  constexpr int NT = NW * 64;
  constexpr int H = K - 1;
  constexpr int HX = ((H + VEC - 1) / VEC) * VEC;
  constexpr int LN = LdsShape<NT, K>::N;
  // LH (see PDK_LH): a second barrier per step separates the exchange of stage 1's inputs
  // from that of the later stages' -- the arrays then need no second copy by step parity,
  // which is what makes room for the stage-0 arrays (108 KB instead of 144)
  // (float, whole-vector rows: the double and the ragged-row instantiations would spill)
  constexpr bool LH = PDK_LH && K == 3 && NW == 12 && !PF2 && sizeof(T) == 4 && !RAG;
  constexpr int NB = LH ? 1 : 2;
  __shared__ __attribute__((aligned(16))) T s_xb[NB][K - 1][LN * VEC];
  __shared__ __attribute__((aligned(16))) T s_py[NB][K - 1][LN * VEC];
  __shared__ T s_px[NB][K - 1][LN];   // last p_x^(k-1) of every lane's vector
  __shared__ __attribute__((aligned(16))) T s_xb0[LH ? LN * VEC : VEC];   // old xbar[s]
  __shared__ __attribute__((aligned(16))) T s_py0[LH ? LN * VEC : VEC];   // old p_y[s]
  __shared__ T s_px0[LH ? LN : 1];                                        // old p_x[s], last

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
  // lane -> (row, lx) of the footprint.  Plain: row-major.  Split (the interior of
  // a row is a multiple of 32 lanes): whole (half) waves hold the interior lanes of
  // one row -- aligned, contiguous 512 B / 1 KiB per array -- and the halo lanes of
  // all rows sit together behind them; rows that straddle waves cost ~15 % of the
  // achievable bandwidth (tools/micro/copy_pattern.hip).
  const int lxb = Q.lxb;
  constexpr int HL = HX / VEC;                 // halo lanes on each side of a row
  int row, lx;
  bool lone = false;                           // neighbours along x are not adjacent lanes
  if (Q.split) {
    const int inner = lxb - 2 * HL;
    const int main = Q.rows * inner;
    if (tid < main) {
      row = tid / inner;
      lx = HL + (tid - row * inner);
      // the interior's first / last lane: the neighbour is a halo lane elsewhere
      lone = (lx == HL) || (lx == HL + inner - 1);
    } else {
      const int h = tid - main;
      row = h / (2 * HL);
      const int side = h - row * (2 * HL);
      lx = side < HL ? side : inner + side;
      lone = true;
    }
  } else {
    row = tid / lxb;
    lx = tid - row * lxb;
  }
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
  // (LH: the same values from lane masks in scalar registers at every use, see
  // masked_scalar)
  const uint64_t m_rin = __builtin_amdgcn_ballot_w64(rin);
  auto sigm = [&](int k) -> T {
    if constexpr (LH) return masked_scalar(S.sigma[k], m_rin); else return sig_m[k];
  };
  auto taum = [&](int k) -> T {
    if constexpr (LH) return masked_scalar(S.tau[k], m_rin); else return tau_m[k];
  };

  // plane indices fit in 32 bits (the host checks nz < 2^30): scalar registers
  // are the scarce resource here (11 buffer descriptors)
  const int nzi = (int)G.nz;
  const int zbeg = zc * zchunk;
  int zend = zbeg + zchunk;
  if (zend > nzi) zend = nzi;
  const int s_first = zbeg > H ? zbeg - H : 0;
  const int s_last = zend + K - 2;
  const int s_hi = s_last < nzi - 1 ? s_last : nzi - 1;   // last step with a stage 1

  // ---- buffer resources over the planes [pz0, pe) this workgroup touches
  const int pz0 = s_first > 0 ? s_first - 1 : 0;
  int pe = s_last + 2;
  if (pe > nzi) pe = nzi;
  const uint32_t szb = (uint32_t)(G.sz * (int64_t)sizeof(T));
  const uint32_t syb = (uint32_t)(G.sy * (int64_t)sizeof(T));
  const uint32_t span = (uint32_t)(pe - pz0) * szb;       // < 2^30 (host-checked)
  const int64_t boff = (int64_t)pz0 * G.sz;
  const uint32_t pspan = S.has_p ? span : 0u;             // p = 0: every load returns 0
  const rsrc_t r_xb = make_rsrc(xbar_in + boff, span);
  const rsrc_t r_x = make_rsrc(x_in + boff, span);
  const rsrc_t r_bt = make_rsrc(bt + boff, span);
  const rsrc_t r_px = make_rsrc(p_in + boff, pspan);
  const rsrc_t r_py = make_rsrc(p_in + G.n + boff, pspan);
  const rsrc_t r_pz = make_rsrc(p_in + 2 * G.n + boff, pspan);
  const rsrc_t w_xb = make_rsrc(xbar_out + boff, span);
  const rsrc_t w_x = make_rsrc(x_out + boff, span);
  const rsrc_t w_px = make_rsrc(p_out + boff, span);
  const rsrc_t w_py = make_rsrc(p_out + G.n + boff, span);
  const rsrc_t w_pz = make_rsrc(p_out + 2 * G.n + boff, span);

  // per-lane byte offsets inside a plane (loop invariant); the plane is selected
  // by the scalar offset.  Stage-1 halos: old data from global memory (L1/L2).
  const bool row_end = (lx == lxb - 1) || lane == 63 || lone;
  const bool row_beg = (lx == 0) || lane == 0 || lone;
  const uint32_t o0 = (uint32_t)((y * G.sy + x0) * (int64_t)sizeof(T));
  const uint32_t v_own = rin ? o0 : kInvalid;
  // (LH: only the footprint's first / last row and first / last lane load their halo;
  // everybody else's comes out of the LDS)
  const bool e_up = !LH || row == 0, e_down = !LH || row == Q.rows - 1;
  const bool e_right = LH ? lx == lxb - 1 : row_end, e_left = LH ? lx == 0 : row_beg;
  uint32_t v_up = (rin && y > 0 && e_up) ? o0 - syb : kInvalid;
  const uint32_t v_down = (rin && y + 1 < G.ny && e_down) ? o0 + syb : kInvalid;
  uint32_t v_right = (rin && e_right && x0 + VEC < G.nx) ? o0 + 16u : kInvalid;
  const uint32_t v_left = (rin && e_left && x0 > 0) ? o0 - (uint32_t)sizeof(T) : kInvalid;
  // (LH: a lane sits in at most one outermost row and is at most one outermost lane, so
  // the row beyond the footprint -- above OR below -- lands in one register vector and
  // the x neighbour beyond it -- right OR left -- in one register: four vector registers
  // and a load less per lane and step)
  const bool last_row = row == Q.rows - 1;
  if (LH && last_row && row != 0) v_up = v_down;
  if (LH && lx == 0 && lxb > 1) v_right = v_left;
  const uint32_t v_pyup = (LH && row != 0) ? kInvalid : v_up;
  const uint32_t v_st = rvalid ? o0 : kInvalid;
  // RAG: rows are not a multiple of the vector width, so the row's last vector
  // sticks out by VEC - nval elements (they belong to the next row).  Their xbar
  // must read as zero -- the zero padding of K -- wherever a valid neighbour looks
  // at it, and they are left out of the stores; whatever else is computed for
  // them is read by nobody.  (Accesses are 4-byte aligned then: legal, 90 % of the
  // aligned rate, tools/micro/unaligned_b128.hip.)
  int nval = VEC;
  if constexpr (RAG) {
    const int64_t left = G.nx - x0;
    nval = left >= VEC ? VEC : (left > 0 ? (int)left : 0);
  }
  auto cut_tail = [&](T (&v)[VEC]) {
    if constexpr (RAG) {
#pragma unroll
      for (int j = 1; j < VEC; ++j) v[j] = j < nval ? v[j] : T(0);
    }
  };
  auto store_vec = [&](rsrc_t r, uint32_t vo, const T (&v)[VEC]) {
    if constexpr (RAG) {
      // (rows held at a pitch: what sticks out of the row is padding -- whole store)
      if (nval < VEC && !G.padded) {
        bst_part<T, VEC>(r, vo, v, nval);
        return;
      }
    }
    bst<T, VEC>(r, vo, v);
  };
  // stages >= 2: neighbours inside the footprint come from LDS; beyond it the
  // value is zero (the zero padding of K / K^T at the volume edge) or belongs
  // to a voxel whose result is recomputed by the next footprint and not stored
  const bool n_r = active && lx < lxb - 1;
  const bool v_l = rin && lx > 0 && x0 > 0;      // the left / upper voxel is in the volume
  const bool v_u = rin && row > 0 && y > 0;
  const bool g_l = rin && row_beg && x0 > 0;
  const bool g_u = rin && y > 0;
  // LDS slot of this lane: logical position, one padding row above.  Idle lanes
  // (row >= rows) keep the plain mapping: their zeros land in the bottom padding.
  const int slot = active ? (row + 1) * lxb + lx : tid + lxb;
  const int li = slot * VEC;
  // step size of the recomputed dual of the voxel above: zero where there is none
  // (its inputs are zero there -- out-of-range load, zero LDS row, or a lane outside
  // the volume -- so the result is exactly zero without a select per value)
  T sig_u[K];
  sig_u[0] = g_u ? S.sigma[0] : T(0);
#pragma unroll
  for (int k = 1; k < K; ++k) sig_u[k] = v_u ? S.sigma[k] : T(0);
  const uint64_t m_gu = __builtin_amdgcn_ballot_w64(g_u), m_vu = __builtin_amdgcn_ballot_w64(v_u);
  auto sigu = [&](int k) -> T {
    if constexpr (LH) return masked_scalar(S.sigma[k], k == 0 ? m_gu : m_vu);
    else return sig_u[k];
  };
  // Stage k is exact -- and its result used -- only k-1 rows inside the footprint
  // (and, for the last stage, off the x halo lanes): the outer rows run stage 1
  // only, the next ones stages 1-2, ...  A wave none of whose lanes needs a stage
  // skips its arithmetic (it still publishes its previous stage and joins the
  // barrier); with the split lane mapping a wave is one row, so 6 of 33
  // row-stages disappear at 11 rows.
  bool wneed[K];
  wneed[0] = true;
#pragma unroll
  for (int k = 2; k <= K; ++k) {
    const bool in_x = single_x || k < K || (lx >= HL && lx < lxb - HL);
    const bool mine = active && row >= k - 1 && row <= Q.rows - k && in_x;
    wneed[k - 1] = __ballot(mine) != 0ull;
  }

  // zero rows above and below the footprint, once
  for (int i = tid; i < NB * (K - 1) * 2 * lxb * VEC; i += NT) {
    const int e = i % (lxb * VEC);
    int r = i / (lxb * VEC);
    const int side = r & 1; r >>= 1;
    const int kk = r % (K - 1);
    const int b = r / (K - 1);
    const int at = side ? (Q.rows + 1) * lxb * VEC + e : e;
    s_xb[b][kk][at] = T(0);
    s_py[b][kk][at] = T(0);
    if (LH && b == 0 && kk == 0) { s_xb0[at] = T(0); s_py0[at] = T(0); }
  }

  // State carried from one plane step to the next.  The 8-wave variants keep two
  // copies: a step reads one and writes the other (main loop unrolled by two), so
  // "carrying" is a renaming of registers instead of ~40 v_mov per step; with 12
  // or 16 waves that spills, and one copy is updated in place.
  struct Carry {
    T xc[VEC];              // xbar[s]
    T pz[K][VEC];           // pz[k-1]: p^(k)_z on the plane stage k finished last
    T c_xb[K][VEC];         // [k-1], k>=2: xbar^(k-1) on plane s-(k-1)
    T c_x[K][VEC];          //               x^(k-1) there
    T c_bt[K][VEC];         //               bt there
    T c_kt[K][VEC];         //               in-plane part of K^T p^(k) there
    T c_px[K][VEC];         // [k-1], k>=3: p_x^(k-1) on plane s-(k-1)+1 (from IP_{k-1})
    T c_py[K][VEC];
  };
  Carry CA, CB;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    zero(CA.pz[k]); zero(CA.c_xb[k]); zero(CA.c_x[k]); zero(CA.c_bt[k]);
    zero(CA.c_kt[k]); zero(CA.c_px[k]); zero(CA.c_py[k]);
  }
  uint32_t adv = (uint32_t)(s_first - pz0) * szb;   // scalar offset of plane s
  bld<T, VEC>(r_xb, v_own, adv, CA.xc);
  cut_tail(CA.xc);
  {
    // p'_z of the plane below the first one (zero at the bottom of the volume)
    T xm[VEC], pm[VEC];
    const uint32_t vb = s_first > 0 ? v_own : kInvalid;
    const uint32_t ab = s_first > 0 ? adv - szb : 0u;
    bld<T, VEC>(r_xb, vb, ab, xm);
    bld<T, VEC>(r_pz, vb, ab, pm);
    const T sg = s_first > 0 ? S.sigma[0] : T(0);
#pragma unroll
    for (int j = 0; j < VEC; ++j)
      CA.pz[0][j] = dual_update_u<HUBER, UNIT>(pm[j], CA.xc[j], xm[j], G.wz, sg, S.hden[0]);
  }

  // Software pipeline: the loads of plane s+1 are issued right after the stage-1
  // arithmetic of plane s and stay in flight while the later stages (and the
  // barrier) run -- the waves of a workgroup move in lockstep, so nothing else
  // would hide the latency.
  // PF2: two register sets, so the loads of plane s+1 are issued BEFORE the
  // stage-1 arithmetic of plane s (a full step of latency hiding; +36 VGPRs).
  typedef PlaneLoads<T, VEC> Loads;
  Loads LA, LB;
  // stage 1's halo of plane `a`
  auto issue_halo = [&](Loads &L, uint32_t a) {
#if PDK_ABLATE & 32
    // (timing experiment, wrong results: stage 1 without its halo loads -- what an
    // exchange of the old xbar / p_y rows inside the workgroup could save at most)
    L.xright = L.xleft = L.pxleft = T(0);
    zero(L.xdown); zero(L.xup); zero(L.pyup);
#else
    if constexpr (LH) {
      // The rows beyond the footprint (every wave issues the two loads, nine of the
      // twelve with every lane out of range: a wave-uniform branch around them cost more
      // than it saved -- 1.250 against 1.168 ms per launch, without the exchange 1.193).
      // The VOXELS beyond the footprint's first / last lane are not loaded at all: they
      // only reach stage 1 of the footprint's outermost voxel, and with HX >= K what a
      // stored value depends on ends one voxel short of it (static_assert below).
      static_assert(!LH || HX >= K, "LH: the x halo must be wider than the dependency cone");
      L.xright = L.pxleft = T(0);
      bld<T, VEC>(r_xb, v_up, a, L.xup);
      bld<T, VEC>(r_py, v_pyup, a, L.pyup);
    } else {
      L.xright = bld1<T>(r_xb, v_right, a);
      L.xleft = bld1<T>(r_xb, v_left, a);
      L.pxleft = bld1<T>(r_px, v_left, a);
      bld<T, VEC>(r_xb, v_down, a, L.xdown);
      bld<T, VEC>(r_xb, v_up, a, L.xup);
      bld<T, VEC>(r_py, v_up, a, L.pyup);
    }
#endif
  };
  auto issue_loads = [&](Loads &L, int sp, uint32_t a) {
    // sp: plane; a: its scalar offset.  The plane above the volume is zero.
#if PDK_PRIO & 1
    __builtin_amdgcn_s_setprio(3);
#endif
    // (LH: the halo loads first.  They are what the next step publishes first, under
    // lane masks, and the compiler waits for the LAST load of a step with vmcnt(0) --
    // the stores issued behind it drained at the top of every step; the own planes'
    // loads it counts)
    if constexpr (LH) issue_halo(L, a);
    bld<T, VEC>(r_xb, (sp + 1 < nzi) ? v_own : kInvalid, a + szb, L.xn);
    bld<T, VEC>(r_x, v_own, a, L.xv);
    bld<T, VEC>(r_bt, v_own, a, L.btn);
    bld<T, VEC>(r_px, v_own, a, L.pxo);
    bld<T, VEC>(r_py, v_own, a, L.pyo);
    bld<T, VEC>(r_pz, v_own, a, L.pzo);
    if constexpr (!LH) issue_halo(L, a);
#if PDK_PRIO & 1
    __builtin_amdgcn_s_setprio(0);
#endif
  };
  issue_loads(LA, s_first, adv);

  // one plane step; HAVE1 = false in the drain steps above the volume's last plane
  auto step = [&](auto have1_tag, int s, Loads &L, Loads &LN, const Carry &P,
                  Carry &N) {
    constexpr bool HAVE1 = decltype(have1_tag)::value;
#if PDK_ABLATE & 8
    {
      // Timing experiment (wrong results): the step's memory traffic without its
      // arithmetic -- the prefetched plane is summed up (every load stays alive),
      // the sum goes through the LDS exchange and the barrier of a step (bit 16:
      // neither) and out through the five stores.
      const bool more = s + 1 <= s_hi;
      T acc[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j)
        acc[j] = L.xn[j] + L.xv[j] + L.btn[j] + L.pxo[j] + L.pyo[j] + L.pzo[j] + L.xdown[j] +
                 L.xup[j] + L.pyup[j] + L.xright + L.xleft + L.pxleft + P.xc[j];
      if constexpr (HAVE1) issue_loads(L, more ? s + 1 : s, more ? adv + szb : adv);
      const int buf = LH ? 0 : (int)(s & 1);
#if !(PDK_ABLATE & 16)
#pragma unroll
      for (int k = 2; k <= K; ++k) {
        stv<T, VEC>(&s_xb[buf][k - 2][li], acc);
        stv<T, VEC>(&s_py[buf][k - 2][li], acc);
        s_px[buf][k - 2][slot] = acc[0];
      }
      __syncthreads();
#pragma unroll
      for (int k = 2; k <= K; ++k) {
        T below[VEC], above[VEC], above_py[VEC];
        ldv<T, VEC>(&s_xb[buf][k - 2][li + lxb * VEC], below);
        ldv<T, VEC>(&s_xb[buf][k - 2][li - lxb * VEC], above);
        ldv<T, VEC>(&s_py[buf][k - 2][li - lxb * VEC], above_py);
        const T e = s_xb[buf][k - 2][li + VEC] + s_xb[buf][k - 2][li - 1] +
                    s_px[buf][k - 2][slot - 1];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += below[j] + above[j] + above_py[j] + e;
      }
#endif
      const int f = s - (K - 1);
      if (f >= zbeg && f < zend) {
        const uint32_t vo = v_st + (adv - (uint32_t)(K - 1) * szb);
        store_vec(w_pz, vo, acc);
        store_vec(w_x, vo, acc);
        store_vec(w_xb, vo, acc);
      }
      const int a = s - (K - 2);
      if (a >= zbeg && a < zend) {
        const uint32_t vo = v_st + (adv - (uint32_t)(K - 2) * szb);
        store_vec(w_px, vo, acc);
        store_vec(w_py, vo, acc);
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) N.xc[j] = acc[j];
      adv += szb;
      return;
    }
#endif
    // fr_*[k-1]: results of stage k produced in this step
    T fr_xb[K][VEC], fr_x[K][VEC], fr_bt[K][VEC], pzn[K][VEC];
    T f_px[VEC], f_py[VEC];
    if constexpr (HAVE1) {
      // prefetch plane s+1 (after the last plane: re-reads it, unused)
      const bool more = s + 1 <= s_hi;
      if constexpr (PF2) issue_loads(LN, more ? s + 1 : s, more ? adv + szb : adv);
#if PDK_ABLATE & 64
      {
        // (timing experiment, with bit 32: what an exchange of stage 1's halo through
        // the LDS would cost -- two vectors and a scalar published, a SECOND barrier in
        // the step, three vectors and three scalars read back)
        const int b0 = LH ? 0 : (int)(s & 1);
        stv<T, VEC>(&s_xb[b0][0][li], P.xc);
        stv<T, VEC>(&s_py[b0][0][li], L.pyo);
        s_px[b0][0][slot] = L.pxo[VEC - 1];
        __syncthreads();
        ldv<T, VEC>(&s_xb[b0][0][li + lxb * VEC], L.xdown);
        ldv<T, VEC>(&s_xb[b0][0][li - lxb * VEC], L.xup);
        ldv<T, VEC>(&s_py[b0][0][li - lxb * VEC], L.pyup);
        L.xright = s_xb[b0][0][li + VEC];
        L.xleft = s_xb[b0][0][li - 1];
        L.pxleft = s_px[b0][0][slot - 1];
      }
#endif
      // ================= stage 1: iteration n+1 on plane s ===================
      // its halo: loaded (every lane its own), or -- LH -- the old values every lane
      // holds anyway exchanged through the LDS: own xbar[s], p_y[s] and the last p_x[s]
      // published, the footprint's outermost rows also put the row they loaded from
      // beyond it into the padding row, a barrier, the neighbours read
      T h_down[VEC], h_up[VEC], h_pyup[VEC];
      T h_right = L.xright, h_left = LH ? L.xright : L.xleft, h_pxleft = L.pxleft;
      if constexpr (LH) {
        __builtin_amdgcn_sched_barrier(0);
        if (active) {
          stv<T, VEC>(&s_xb0[li], P.xc);
          stv<T, VEC>(&s_py0[li], L.pyo);
          s_px0[slot] = L.pxo[VEC - 1];
          if (row == 0) {
            stv<T, VEC>(&s_xb0[li - lxb * VEC], L.xup);
            stv<T, VEC>(&s_py0[li - lxb * VEC], L.pyup);
          }
          // (the last row's L.xup holds the row BELOW the footprint; a one-row footprint
          // would need both and does not exist: rows >= 2 K - 1)
          if (last_row && row != 0) stv<T, VEC>(&s_xb0[li + lxb * VEC], L.xup);
        }
        __syncthreads();
        ldv<T, VEC>(&s_xb0[li + lxb * VEC], h_down);
        ldv<T, VEC>(&s_xb0[li - lxb * VEC], h_up);
        ldv<T, VEC>(&s_py0[li - lxb * VEC], h_pyup);
        if (lx != lxb - 1) h_right = s_xb0[li + VEC];
        if (lx != 0) {
          h_left = s_xb0[li - 1];
          h_pxleft = s_px0[slot - 1];
        }
        // (the halo-free parts of stage 1 stay below the exchange: hoisted above the
        // barrier they hold their results across it and the kernel spills)
        __builtin_amdgcn_sched_barrier(0);
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          h_down[j] = L.xdown[j];
          h_up[j] = L.xup[j];
          h_pyup[j] = L.pyup[j];
        }
      }
      T nb = __shfl_down(P.xc[0], 1, kWave);
      if (row_end) nb = h_right;
      const T sm0 = sigm(0), tm0 = taum(0), su0 = sigu(0);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const T hx = (j + 1 < VEC) ? P.xc[(j + 1) % VEC] : nb;
        f_px[j] = dual_update_u<HUBER, UNIT>(L.pxo[j], hx, P.xc[j], G.wx, sm0, S.hden[0]);
      }
      dual_vec<HUBER, UNIT>(f_py, L.pyo, h_down, P.xc, G.wy, sm0, S.hden[0]);
      dual_vec<HUBER, UNIT>(pzn[0], L.pzo, L.xn, P.xc, G.wz, sm0, S.hden[0]);
      T pxl = __shfl_up(f_px[VEC - 1], 1, kWave);
      if (row_beg)
        pxl = g_l ? dual_update_u<HUBER, UNIT>(h_pxleft, P.xc[0], h_left, G.wx, S.sigma[0],
                                         S.hden[0])
                  : T(0);
      // the upper neighbour's new dual; without one, pyup = xup = 0 (offset out
      // of range) and sig_u = 0 make it exactly zero
      T puv[VEC];
      dual_vec<HUBER, UNIT>(puv, h_pyup, P.xc, h_up, G.wy, su0, S.hden[0]);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const T pu = puv[j];
        const T pl = (j > 0) ? f_px[(j + VEC - 1) % VEC] : pxl;
        T kt = adj_term<UNIT>(f_px[j], pl, G.wx);
        kt += adj_term<UNIT>(f_py[j], pu, G.wy);
        kt += adj_term<UNIT>(pzn[0][j], P.pz[0][j], G.wz);
        const T u = L.xv[j] - tm0 * kt;
        const T xnew = prox_data_s<L1>(u, L.btn[j], S.tl[0], S.optl[0]);
        fr_x[0][j] = xnew;
        fr_xb[0][j] = xnew + S.theta[0] * (xnew - L.xv[j]);
        fr_bt[0][j] = L.btn[j];
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) N.xc[j] = L.xn[j];
      cut_tail(N.xc);
      if constexpr (K > 1) cut_tail(fr_xb[0]);
      if constexpr (!PF2) issue_loads(L, more ? s + 1 : s, more ? adv + szb : adv);
    } else {
      // (LH: the arrays of the later stages have one copy -- the barrier that stage 1's
      // exchange brings in a full step separates this step's writes from the reads of
      // the step before here too)
      if constexpr (LH) __syncthreads();
      zero(fr_xb[0]); zero(fr_x[0]); zero(fr_bt[0]); zero(pzn[0]);
      zero(f_px); zero(f_py);
#pragma unroll
      for (int j = 0; j < VEC; ++j) N.xc[j] = P.xc[j];
    }

    // ================= F_k: finish iteration n+k on plane s-(k-1) ===========
#pragma unroll
    for (int k = 2; k <= K; ++k) {
      const int f = s - (k - 1);
      // below the volume and above it everything is zero padding
      const bool inr = f >= 0 && f < nzi;
      // the last stage needs no per-lane masks: what it computes for a voxel
      // outside the volume is neither stored nor read by anyone
      // (wneed first: sigm / taum are formed by volatile instructions)
      if (!wneed[k - 1]) {                         // wave-uniform
        zero(fr_x[k - 1]); zero(fr_xb[k - 1]); zero(fr_bt[k - 1]); zero(pzn[k - 1]);
        continue;
      }
      const T sk = inr ? (k == K ? S.sigma[K - 1] : sigm(k - 1)) : T(0);
      const T tk = inr ? (k == K ? S.tau[K - 1] : taum(k - 1)) : T(0);
      T pkz[VEC];
      dual_vec<HUBER, UNIT>(pkz, P.pz[k - 2], fr_xb[k - 2], P.c_xb[k - 1], G.wz, sk,
                            S.hden[k - 1]);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        T kt = P.c_kt[k - 1][j];
        kt += adj_term<UNIT>(pkz[j], P.pz[k - 1][j], G.wz);
        const T u = P.c_x[k - 1][j] - tk * kt;
        const T xk = prox_data_s<L1>(u, P.c_bt[k - 1][j], S.tl[k - 1], S.optl[k - 1]);
        fr_x[k - 1][j] = xk;
        fr_xb[k - 1][j] = xk + S.theta[k - 1] * (xk - P.c_x[k - 1][j]);
        fr_bt[k - 1][j] = P.c_bt[k - 1][j];
        pzn[k - 1][j] = pkz[j];
      }
      if (k < K) cut_tail(fr_xb[k - 1]);
      if (k == K && f >= zbeg && f < zend) {      // uniform
        const uint32_t vo = v_st + (adv - (uint32_t)(K - 1) * szb);
        store_vec(w_pz, vo, pkz);
        store_vec(w_x, vo, fr_x[K - 1]);
        store_vec(w_xb, vo, fr_xb[K - 1]);
      }
    }

    // ================= IP_k: in-plane part of iteration n+k on plane s-(k-2) =
    const int buf = LH ? 0 : (int)(s & 1);
#pragma unroll
    for (int k = 2; k <= K; ++k) {
      stv<T, VEC>(&s_xb[buf][k - 2][li], fr_xb[k - 2]);
      if (k == 2) {
        stv<T, VEC>(&s_py[buf][0][li], f_py);
        s_px[buf][0][slot] = f_px[VEC - 1];
      } else {
        stv<T, VEC>(&s_py[buf][k - 2][li], P.c_py[k - 1]);
        s_px[buf][k - 2][slot] = P.c_px[k - 1][VEC - 1];
      }
    }
#if !(PDK_ABLATE & 4)
    __syncthreads();
#endif
    T n_kt[K][VEC], n_px[K][VEC], n_py[K][VEC];
#pragma unroll
    for (int k = 2; k <= K; ++k) {
      if (!wneed[k - 1]) {                         // wave-uniform
        zero(n_kt[k - 1]); zero(n_px[k - 1]); zero(n_py[k - 1]);
        continue;
      }
      T below[VEC], above[VEC], above_py[VEC];
      ldv<T, VEC>(&s_xb[buf][k - 2][li + lxb * VEC], below);
      ldv<T, VEC>(&s_xb[buf][k - 2][li - lxb * VEC], above);
      ldv<T, VEC>(&s_py[buf][k - 2][li - lxb * VEC], above_py);
      T right = s_xb[buf][k - 2][li + VEC];
      if (!n_r) right = T(0);
      const T left_xb = s_xb[buf][k - 2][li - 1];
      const T left_px = s_px[buf][k - 2][slot - 1];
      const T pl0 = v_l ? dual_update_u<HUBER, UNIT>(left_px, fr_xb[k - 2][0], left_xb, G.wx,
                                               S.sigma[k - 1], S.hden[k - 1])
                        : T(0);
      T pkx[VEC], pky[VEC], puv[VEC];
      const T sgk = (k == K) ? S.sigma[K - 1] : sigm(k - 1);    // see F_k
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const T px_old = (k == 2) ? f_px[j] : P.c_px[k - 1][j];
        const T hx = (j + 1 < VEC) ? fr_xb[k - 2][(j + 1) % VEC] : right;
        pkx[j] = dual_update_u<HUBER, UNIT>(px_old, hx, fr_xb[k - 2][j], G.wx, sgk,
                                      S.hden[k - 1]);
      }
      if (k == 2)
        dual_vec<HUBER, UNIT>(pky, f_py, below, fr_xb[k - 2], G.wy, sgk, S.hden[k - 1]);
      else
        dual_vec<HUBER, UNIT>(pky, P.c_py[k - 1], below, fr_xb[k - 2], G.wy, sgk,
                              S.hden[k - 1]);
      dual_vec<HUBER, UNIT>(puv, above_py, fr_xb[k - 2], above, G.wy, sigu(k - 1),
                            S.hden[k - 1]);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const T pu = puv[j];
        const T pl = (j > 0) ? pkx[(j + VEC - 1) % VEC] : pl0;
        T kt = adj_term<UNIT>(pkx[j], pl, G.wx);
        kt += adj_term<UNIT>(pky[j], pu, G.wy);
        n_kt[k - 1][j] = kt;
        n_px[k - 1][j] = pkx[j];
        n_py[k - 1][j] = pky[j];
      }
      if (k == K) {
        const int a = s - (K - 2);
        if (a >= zbeg && a < zend) {              // uniform
          const uint32_t vo = v_st + (adv - (uint32_t)(K - 2) * szb);
          store_vec(w_px, vo, pkx);
          store_vec(w_py, vo, pky);
        }
      }
    }

    // ================= next state ==========================================
#pragma unroll
    for (int k = K; k >= 2; --k) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        N.c_xb[k - 1][j] = fr_xb[k - 2][j];
        N.c_x[k - 1][j] = fr_x[k - 2][j];
        N.c_bt[k - 1][j] = fr_bt[k - 2][j];
        N.c_kt[k - 1][j] = n_kt[k - 1][j];
        if (k < K) {
          N.c_px[k][j] = n_px[k - 1][j];
          N.c_py[k][j] = n_py[k - 1][j];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int j = 0; j < VEC; ++j) N.pz[k][j] = pzn[k][j];
    adv += szb;
  };

  __syncthreads();   // LDS padding rows are zero before anyone reads them
  int s = s_first;
  if constexpr (PF2) {
    // 8 waves: registers to spare -> two load sets and two state sets
    for (; s + 1 <= s_hi; s += 2) {
      step(std::true_type{}, s, LA, LB, CA, CB);
      step(std::true_type{}, s + 1, LB, LA, CB, CA);
    }
    if (s <= s_hi) {
      step(std::true_type{}, s, LA, LB, CA, CB);
      CA = CB;
      ++s;
    }
  } else {
    // one set of each, updated in place (a step writes its state last)
    for (; s <= s_hi; ++s) step(std::true_type{}, s, LA, LA, CA, CA);
  }
  for (; s <= s_last; ++s)       // at most K-1 drain steps
    step(std::false_type{}, s, LA, LA, CA, CA);
}

struct Tuning {
  int enable = 1;
  int kmax = 3;
  int nw = 0;         // 0 = choose (12 or 8 waves for K = 3; 16 / 12 for K = 2)
  int zchunk = 0;     // 0 = choose
  int ntx = 0;        // 0 = choose
  int xcd_map = 1;
  int verbose = 0;
  int autotune = 1;   // time the best few model candidates once per problem shape
  int tune_min_mvox = 16;   // ... for volumes of at least this many Mi voxels
  int pf2 = -1;       // -1 = where the registers allow; 0 / 1 force
  int tail2 = 1;        // a run's trailing pair through depth 2 of this kernel once the
                        // shape's depth-3 plan has settled (else k_pd_fused2)
  int min_kvox = 1024;  // smaller volumes (Ki voxels) stay with the one-iteration kernel:
                        // cache resident, they want many short workgroups (crossover
                        // measured between 96^3 and 128^3, tools/crossover_pd.py)
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

int g_split = -1;   // -1 / 1: split lane mapping where it applies; 0: never (experiment)

struct Config {
  int nw = 0;
  int pf2 = 0;       // 1 = two prefetch register sets
  Tiling q{};
  int64_t zchunk = 0;
  double cost = 0.0;      // modelled bytes through L2 -> fabric per launch
};

// Footprint of `nt` lanes cut into ntx columns of tiles.
// xv_fixed > 0: that many valid voxels per tile (the last tile may be shorter)
// instead of nx spread evenly over ntx tiles
inline bool make_tiling(int64_t nx, int64_t ny, int nt, int vec, int h, int hx,
                        int lxm, int ntx, Tiling *out, int64_t xv_fixed = 0) {
  Tiling q;
  if (ntx == 1) {
    q.xv = 0;
    q.lxb = (int)((nx + vec - 1) / vec);
  } else {
    int64_t xv = xv_fixed > 0 ? xv_fixed : (nx + ntx - 1) / ntx;
    xv = (xv + vec - 1) / vec * vec;
    if (xv * (ntx - 1) >= nx) return false;   // fewer tiles would do
    q.xv = (int)xv;
    q.lxb = (int)(xv / vec) + 2 * (hx / vec);
  }
  if (q.lxb < 1 || q.lxb > lxm) return false;
  q.rows = nt / q.lxb;
  if (q.rows - 2 * h < 1) return false;
  q.ntx = ntx;
  q.nty = (int)((ny + (q.rows - 2 * h) - 1) / (q.rows - 2 * h));
  // interior of a row = whole half-waves: give them to whole half-waves
  q.split = (ntx > 1 && (q.xv / vec) % 32 == 0 && g_split != 0) ? 1 : 0;
  *out = q;
  return true;
}

// z-chunk length: trade the 2(K-1) extra planes per chunk against filling the
// last round of workgroups (one workgroup per CU).  Returns the efficiency.
inline double pick_zchunk(int64_t nz, int64_t tiles, int extra, int64_t max_chunk,
                          int64_t *chunk_out) {
  const double slots = (double)cu_count();
  double best = -1.0;
  int64_t best_chunk = nz < max_chunk ? nz : max_chunk;
  for (int64_t nzc = 1; nzc <= nz; ++nzc) {
    const int64_t chunk = (nz + nzc - 1) / nzc;
    if (chunk < 8 && nzc > 1) break;
    if (chunk > max_chunk) continue;
    if ((nz + chunk - 1) / chunk != nzc) continue;
    const double blocks = (double)tiles * nzc;
    const double rounds = (double)(int64_t)((blocks + slots - 1) / slots);
    const double fill = blocks / (rounds * slots);
    const double eff = fill * (double)chunk / (double)(chunk + extra);
    if (eff > best) { best = eff; best_chunk = chunk; }
  }
  *chunk_out = best_chunk;
  return best > 0.0 ? best : 1.0;
}

// Bytes one launch pulls through L2 if no overlap is shared between footprints:
// whole 128-byte lines per footprint row (the x halo costs a line on each side,
// which is what makes narrow tiles expensive), six input arrays, plus the five
// output arrays once.
inline double model_cost(const Tiling &q, int64_t nx, int64_t ny, int64_t nz,
                         int vec, int esize, int h, int hx, double zeff) {
  double lines = 0.0;
  for (int tx = 0; tx < q.ntx; ++tx) {
    int64_t lo = q.xv ? (int64_t)tx * q.xv - hx : 0;
    int64_t hi = lo + (int64_t)q.lxb * vec;
    if (lo < 0) lo = 0;
    if (hi > nx) hi = nx;
    if (hi <= lo) continue;
    lines += (double)((hi * esize - 1) / 128 - (lo * esize) / 128 + 1);
  }
  double rows = 0.0;
  const int64_t tyv = q.rows - 2 * h;
  for (int ty = 0; ty < q.nty; ++ty) {
    int64_t lo = (int64_t)ty * tyv - h, hi = lo + q.rows;
    if (lo < 0) lo = 0;
    if (hi > ny) hi = ny;
    if (hi > lo) rows += (double)(hi - lo);
  }
  const double reads = lines * 128.0 * rows * (double)nz * 6.0;
  const double writes = 5.0 * (double)nx * ny * nz * esize;
  return (reads + writes) / zeff;
}

template <typename T>
inline bool al16(const T *a) {
  return (reinterpret_cast<uintptr_t>(a) & 15u) == 0;
}

template <typename T, int VEC, int NW, int K, int WPE, bool HUBER, bool L1, bool PF2,
          bool UNIT, bool RAG>
int launch_f(const Config &c, const T *xbar_in, T *xbar_out, const T *x_in, T *x_out,
             const T *bt, const T *p_in, T *p_out, const Geom<T> &G,
             const StageScalars<T, K> &S, hipStream_t st) {
  const Tiling &Q = c.q;
  const int64_t tiles = (int64_t)Q.ntx * Q.nty;
  const int64_t nzc = (G.nz + c.zchunk - 1) / c.zchunk;
  int64_t blocks = tiles * nzc;
  int64_t slab = 0;
  if (g_tunek.xcd_map && tiles >= 16) {
    slab = (tiles + 7) / 8;
    blocks = 8 * slab * nzc;
  }
  if (blocks > 0x7fffffff) return NSOL_EINVAL;
  if (g_tunek.verbose == 1)
    fprintf(stderr, "k_pd_fusedk K=%d waves=%d pf2=%d split=%d: lxb=%d rows=%d xv=%d tiles=%dx%d "
            "zchunk=%lld blocks=%lld slab=%lld model=%.3f GB\n", K, NW, (int)PF2, Q.split, Q.lxb, Q.rows,
            Q.xv, Q.ntx, Q.nty, (long long)c.zchunk, (long long)blocks,
            (long long)slab, c.cost * 1e-9);
  hipLaunchKernelGGL((k_pd_fusedk<T, VEC, NW, K, WPE, HUBER, L1, PF2, UNIT, RAG>),
                     dim3((unsigned)blocks), dim3(NW * 64), 0, st, xbar_in, xbar_out,
                     x_in, x_out, bt, p_in, p_out, G, S, Q, (int)c.zchunk, (int)slab);
  g_launches[K == 3 ? 1 : 0].fetch_add(1, std::memory_order_relaxed);
  return launch_status();
}

template <typename T, int VEC, int NW, int K, int WPE, bool PF2, bool RAG>
int launch_k(const Config &c, const T *xbar_in, T *xbar_out, const T *x_in, T *x_out,
             const T *bt, const T *p_in, T *p_out, const Geom<T> &G,
             const StageScalars<T, K> &S, int flags, hipStream_t st) {
#define NSOL_F(HB, L)                                                           \
  (unit ? launch_f<T, VEC, NW, K, WPE, HB, L, PF2, true, RAG>(                  \
              c, xbar_in, xbar_out, x_in, x_out, bt, p_in, p_out, G, S, st)     \
        : launch_f<T, VEC, NW, K, WPE, HB, L, PF2, false, RAG>(                 \
              c, xbar_in, xbar_out, x_in, x_out, bt, p_in, p_out, G, S, st))
  const bool unit = G.wx == T(1) && G.wy == T(1) && G.wz == T(1);
  const bool huber = (flags & NSOL_PD_REG_HUBER) != 0;
  const bool l1 = (flags & NSOL_PD_DATA_L1) != 0;
  if (huber) return l1 ? NSOL_F(true, true) : NSOL_F(true, false);
  return l1 ? NSOL_F(false, true) : NSOL_F(false, false);
#undef NSOL_F
}

// workgroup sizes compiled per depth (16 waves: neither the registers nor the
// LDS of K = 3 fit)
template <int K> struct Waves;
template <> struct Waves<2> { static constexpr int n = 3; static constexpr int v[3] = {16, 12, 8}; };
template <> struct Waves<3> { static constexpr int n = 2; static constexpr int v[2] = {12, 8}; };

template <typename T, int K>
int launch_cfg(const Config &c, const T *xbar_in, T *xbar_out, const T *x_in,
               T *x_out, const T *bt, const T *p_in, T *p_out, const Geom<T> &G,
               const StageScalars<T, K> &S, int flags, hipStream_t st) {
  constexpr int VW = 16 / sizeof(T);
#define NSOL_W(NWV, WPE, PF)                                                      \
  launch_k<T, VW, NWV, K, WPE, PF, false>(c, xbar_in, xbar_out, x_in, x_out, bt,    \
                                          p_in, p_out, G, S, flags, st)
  // rows that are not a multiple of the vector width: 12-wave workgroups only
  // (candidates() offers nothing else for such shapes)
  if (G.nx % VW != 0) {
    if (c.nw != 12) return NSOL_EINVAL;
    return launch_k<T, VW, 12, K, 3, false, true>(c, xbar_in, xbar_out, x_in, x_out, bt,
                                                  p_in, p_out, G, S, flags, st);
  }
  // two prefetch register sets where the register file has room for them
  if (c.nw == 8) return c.pf2 ? NSOL_W(8, 2, true) : NSOL_W(8, 2, false);
  if constexpr (K == 2) {
    if (c.nw == 12) return NSOL_W(12, 3, false);
    if (c.nw == 16) return NSOL_W(16, 4, false);
  } else {
    if (c.nw == 12) return NSOL_W(12, 3, false);
  }
#undef NSOL_W
  return NSOL_EINVAL;
}

template <int NT, int K> constexpr int lxm_of() { return LdsShape<NT, K>::LXM; }

// candidate configurations for a shape, cheapest (by model) first
template <typename T, int K>
std::vector<Config> candidates(const Geom<T> &G) {
  constexpr int VW = 16 / sizeof(T);
  constexpr int H = K - 1;
  constexpr int HX = ((H + VW - 1) / VW) * VW;
  std::vector<Config> out;
  const int64_t plane_bytes = G.sz * (int64_t)sizeof(T);
  // 32-bit buffer offsets: the planes one workgroup touches span < 2^30 bytes
  const int64_t max_planes = ((int64_t)1 << 30) / plane_bytes - 1;
  const int64_t max_chunk = max_planes - 2 * K - 1;
  if (max_chunk < 8) return out;
  for (int wi = 0; wi < Waves<K>::n; ++wi) {
    const int nw = Waves<K>::v[wi];
    if (g_tunek.nw > 0 && g_tunek.nw != nw) continue;
    if (G.nx % VW != 0 && nw != 12) continue;     // see launch_cfg
    const int nt = nw * 64;
    const int lxm = nt / (2 * K);
    // tile counts 1..64 with nx spread evenly, plus tiles whose interior is exactly
    // one or two waves wide (256 / 512 B-aligned rows for the split lane mapping)
    // where nx is not a multiple of that
    struct Spec { int ntx; int64_t xv; };
    std::vector<Spec> specs;
    for (int ntx = 1; ntx <= 64; ++ntx) specs.push_back({ntx, 0});
    for (int lanes : {64, 128}) {
      const int64_t xv = (int64_t)lanes * VW;
      if (G.nx > xv && G.nx % xv != 0)
        specs.push_back({(int)((G.nx + xv - 1) / xv), xv});
    }
    for (const Spec &sp : specs) {
      const int ntx = sp.ntx;
      if (g_tunek.ntx > 0 && ntx != g_tunek.ntx) continue;
      Config c;
      c.nw = nw;
      c.pf2 = (nw == 8) ? 1 : 0;
      if (g_tunek.pf2 == 0) c.pf2 = 0;
      if (!make_tiling(G.nx, G.ny, nt, VW, H, HX, lxm, ntx, &c.q, sp.xv)) continue;
      const int64_t tiles = (int64_t)c.q.ntx * c.q.nty;
      double zeff;
      if (g_tunek.zchunk > 0) {
        c.zchunk = g_tunek.zchunk < max_chunk ? g_tunek.zchunk : max_chunk;
        if (c.zchunk > G.nz) c.zchunk = G.nz;
        zeff = (double)c.zchunk / (double)(c.zchunk + 2 * H);
      } else {
        zeff = pick_zchunk(G.nz, tiles, 2 * H, max_chunk, &c.zchunk);
      }
      c.cost = model_cost(c.q, G.nx, G.ny, G.nz, VW, (int)sizeof(T), H, HX, zeff);
      // fewer waves hide less latency: mild penalty so that ties go to more waves
      c.cost *= 1.0 + 0.02 * (16 - nw) / 4.0;
      if (c.q.split) {
        // rows of whole waves stream ~15 % better (measured); rows of half waves
        // gain less and not always: the plain mapping stays a candidate there
        const bool whole = ((c.q.lxb - 2 * (HX / VW)) % 64) == 0;
        if (!whole) {
          Config d = c;
          d.q.split = 0;
          out.push_back(d);
        }
        c.cost *= whole ? 0.88 : 0.97;
      }
      out.push_back(c);
    }
  }
  std::sort(out.begin(), out.end(),
            [](const Config &a, const Config &b) { return a.cost < b.cost; });
  return out;
}

// A plan belongs to one device (its events live there and boxes differ), one
// element size, one depth and one volume shape.  The kernel variants of a shape
// (TV / Huber, l1 / l2, unit / non-unit spacing) share it on purpose: they differ
// in arithmetic per voxel, not in the footprint geometry the plan chooses.
struct PlanKey {
  int device, esize, k;
  int64_t nz, ny, nx;
  int64_t pitch = 0;     // row pitch of a padded layout (0: contiguous rows)
  bool operator<(const PlanKey &o) const {
    return std::tie(device, esize, k, nz, ny, nx, pitch) <
           std::tie(o.device, o.esize, o.k, o.nz, o.ny, o.nx, o.pitch);
  }
};
inline int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  return dev;
}
// Online autotuner.  A plan holds the model's best few configurations; while a
// plan is exploring, every REAL launch of the run uses the next candidate and
// is bracketed by two events that are read back later with hipEventQuery -- no
// trial launches, no host synchronisation.  After `kRounds` samples per
// candidate and `kFinalRounds` for those within 6 % of the best, the finalist
// with the smallest median is kept for the life of the process.
struct Sample { hipEvent_t e0, e1; int cand; };
struct Plan {
  std::vector<Config> cand;
  std::vector<std::vector<float>> samples;
  std::vector<float> best_ms;
  std::vector<int> issued, done;
  std::vector<char> dropped;
  std::deque<Sample> pending;
  int chosen = -1;
};
constexpr int kRounds = 2;
// ... and the finalists -- whoever is within 6 % of the best after those -- are sampled
// up to kFinalRounds times, taking turns: two samples each, minutes apart on a part whose
// clock drifts, settled 511^3 on plans 14 % apart from run to run
constexpr int kFinalRounds = 6;     // (the first sample of a configuration is cold: 5 that count)
constexpr float kFinalBand = 1.06f;
std::map<PlanKey, Plan> g_plans;

// samples candidate i is owed (0: dropped)
inline int plan_target(const Plan &P, int i, float best) {
  if (P.dropped[i]) return 0;
  if (P.done[i] >= kRounds && best > 0.f && P.best_ms[i] > kFinalBand * best) return kRounds;
  return kFinalRounds;
}
inline float plan_best(const Plan &P) {
  float best = -1.f;
  for (int i = 0; i < (int)P.cand.size(); ++i)
    if (!P.dropped[i] && P.done[i] >= 1 && (best < 0.f || P.best_ms[i] < best))
      best = P.best_ms[i];
  return best;
}
std::mutex g_plans_mutex;

inline void plan_poll(Plan &P, int K) {
  while (!P.pending.empty()) {
    Sample sm = P.pending.front();
    hipError_t q = hipEventQuery(sm.e1);
    if (q == hipErrorNotReady) break;
    float ms = 0.f;
    if (q == hipSuccess && hipEventElapsedTime(&ms, sm.e0, sm.e1) == hipSuccess) {
      if (P.best_ms[sm.cand] < 0.f || ms < P.best_ms[sm.cand]) P.best_ms[sm.cand] = ms;
      P.samples[sm.cand].push_back(ms);
      P.done[sm.cand]++;
    } else {
      (void)hipGetLastError();
      P.issued[sm.cand]--;              // lost sample: take it again
    }
    (void)hipEventDestroy(sm.e0);
    (void)hipEventDestroy(sm.e1);
    P.pending.pop_front();
  }
  if (P.chosen >= 0) return;
  const int n = (int)P.cand.size();
  // (nobody is dropped on ONE sample: a single slow launch -- the part's clock, another
  // process -- would bar the best tiling for the life of the process; candidates
  // outside the band simply stop at kRounds samples)
  bool all = true;
  const float best_now = plan_best(P);
  for (int i = 0; i < n; ++i)
    if (P.done[i] < plan_target(P, i, best_now)) all = false;
  if (all) {
    // the finalists by the SECOND SMALLEST of their warm samples -- the first launch of a
    // configuration is always slow (1.2 - 1.5 ms where the others take 1.13), and what
    // disturbs a launch afterwards only ever slows it down: the median of five let two or
    // three disturbed launches decide (a battery settled on 12 : 3 : 256, 1.20 ms, once in
    // three runs); one lucky launch must not decide either.  Everybody else by the minimum
    // of two.
    auto score = [&](int i) {
      if ((int)P.samples[i].size() < kFinalRounds) return P.best_ms[i] * kFinalBand;
      std::vector<float> v(P.samples[i].begin() + 1, P.samples[i].end());
      std::sort(v.begin(), v.end());
      return v[1];
    };
    int arg = 0;
    for (int i = 0; i < n; ++i)
      if (!P.dropped[i] && (P.dropped[arg] || score(i) < score(arg))) arg = i;
    P.chosen = arg;
    if (g_tunek.verbose) {
      for (int i = 0; i < n; ++i)
      {
        fprintf(stderr, "k_pd_fusedk tune K=%d waves=%d ntx=%d zchunk=%lld: %.3f ms%s  [",
                K, P.cand[i].nw, P.cand[i].q.ntx, (long long)P.cand[i].zchunk,
                P.best_ms[i], i == arg ? "  <- kept" : (P.dropped[i] ? "  (dropped)" : ""));
        for (float v : P.samples[i]) fprintf(stderr, " %.3f", v);
        fprintf(stderr, " ]\n");
      }
    }
  }
}

template <typename T, int K>
int fusedk_k(const T *xbar_in, T *xbar_out, const T *x_in, T *x_out, const T *bt,
             const T *p_in, T *p_out, const Geom<T> &G, const double *sigma,
             const double *hden, const double *tau, const double *tl,
             const double *theta, int flags, hipStream_t st) {
  StageScalars<T, K> S;
  for (int i = 0; i < K; ++i) {
    S.sigma[i] = (T)sigma[i]; S.hden[i] = huber_den<T>(hden[i]); S.tau[i] = (T)tau[i];
    S.tl[i] = (T)tl[i]; S.optl[i] = prox_den<T>(tl[i]); S.theta[i] = (T)theta[i];
  }
  S.has_p = p_in != nullptr ? 1 : 0;
#define NSOL_GO(cfg)                                                              \
  launch_cfg<T, K>(cfg, xbar_in, xbar_out, x_in, x_out, bt, p_in, p_out, G, S, flags, st)
  const bool forced = g_tunek.nw > 0 || g_tunek.ntx > 0 || g_tunek.zchunk > 0;
  if (forced) {                     // experiments and tests: no plan, no tuning
    std::vector<Config> cand = candidates<T, K>(G);
    if (cand.empty()) return -2;
    return NSOL_GO(cand[0]);
  }
  const PlanKey key{current_device(), (int)sizeof(T), K, G.nz, G.ny, G.nx,
                    G.padded ? G.sy : 0};
  std::lock_guard<std::mutex> lock(g_plans_mutex);
  auto it = g_plans.find(key);
  if (it == g_plans.end()) {
    Plan P;
    std::vector<Config> cand = candidates<T, K>(G);
    if (cand.empty()) return -2;
    const bool big = G.n >= ((int64_t)g_tunek.tune_min_mvox << 20);
    // Depth 2 only ever runs the trailing pair of a run: ONE launch per run, far too
    // few to explore on.  Where the depth-3 plan of the shape has settled, depth 2
    // does not explore either: it takes 12-wave workgroups on whole rows where a row
    // fits one (what 81 exploring launches settle on at 512^3 -- 12 : 1 : 256, 1.02 ms
    // a pair against 1.19 for k_pd_fused2 and for the depth-3 plan's own tiling,
    // tools/_probe/tail2_tuned.py), else the model's best tiling with the depth-3
    // plan's workgroup size and tile count along x.
    int seeded = -1;
    if (K == 2 && g_tunek.autotune && big) {
      auto it3 = g_plans.find(PlanKey{key.device, key.esize, 3, G.nz, G.ny, G.nx, key.pitch});
      if (it3 != g_plans.end() && it3->second.chosen >= 0) {
        const Config &c3 = it3->second.cand[it3->second.chosen];
        for (int i = 0; i < (int)cand.size() && seeded < 0; ++i)
          if (cand[i].nw == 12 && cand[i].q.ntx == 1) seeded = i;
        for (int i = 0; i < (int)cand.size() && seeded < 0; ++i)
          if (cand[i].nw == c3.nw && cand[i].q.ntx == c3.q.ntx) seeded = i;
      }
    }
    if (seeded >= 0) {
      P.cand.push_back(cand[seeded]);
      P.chosen = 0;
    } else if (g_tunek.autotune && big && cand.size() > 1) {
      // per workgroup size: the tilings with up to 8 tiles along x plus the
      // model's best five, each also with a shorter z-chunk (more workgroups in
      // flight).  The model ranks; the measurement decides.
      int per_nw[17] = {0};
      for (const Config &c : cand) {
        if (per_nw[c.nw]++ >= 5 && c.q.ntx > 8) continue;
        P.cand.push_back(c);
        const int64_t zc = c.zchunk > 96 ? 64 : 24;
        if (zc < c.zchunk && zc <= G.nz) {
          Config d = c;
          d.zchunk = zc;
          P.cand.push_back(d);
        }
      }
    } else {
      P.cand.push_back(cand[0]);
      P.chosen = 0;
    }
    const size_t n = P.cand.size();
    P.best_ms.assign(n, -1.f);
    P.samples.assign(n, std::vector<float>());
    P.issued.assign(n, 0);
    P.done.assign(n, 0);
    P.dropped.assign(n, 0);
    it = g_plans.emplace(key, std::move(P)).first;
  }
  Plan &P = it->second;
  // no event queries / records on a stream that is being captured into a graph:
  // such launches use the settled plan, or the model's best while exploring
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess) (void)hipGetLastError();
  if (cap != hipStreamCaptureStatusNone)
    return NSOL_GO(P.cand[P.chosen >= 0 ? P.chosen : 0]);
  plan_poll(P, K);
  if (P.chosen >= 0) return NSOL_GO(P.cand[P.chosen]);
  // exploring: next candidate that still needs a sample (the first launch of a
  // run reads no dual variable, so it is cheaper than all others: not a sample)
  int pick = -1;
  {
    const float best_now = plan_best(P);
    for (int r = 1; r <= kFinalRounds && pick < 0 && S.has_p; ++r)
      for (int i = 0; i < (int)P.cand.size() && pick < 0; ++i) {
        const int owed = plan_target(P, i, best_now);
        if (P.issued[i] < (r < owed ? r : owed)) pick = i;
      }
  }
  if (pick < 0) {
    // every sample is in flight: run the best one known so far meanwhile
    pick = 0;
    for (int i = 0; i < (int)P.cand.size(); ++i)
      if (!P.dropped[i] && P.best_ms[i] >= 0.f &&
          (P.best_ms[pick] < 0.f || P.best_ms[i] < P.best_ms[pick]))
        pick = i;
    return NSOL_GO(P.cand[pick]);
  }
  Sample sm;
  sm.cand = pick;
  if (hipEventCreate(&sm.e0) != hipSuccess) { (void)hipGetLastError(); return NSOL_GO(P.cand[pick]); }
  if (hipEventCreate(&sm.e1) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipEventDestroy(sm.e0);
    return NSOL_GO(P.cand[pick]);
  }
  (void)hipEventRecord(sm.e0, st);
  const int rc = NSOL_GO(P.cand[pick]);
  (void)hipEventRecord(sm.e1, st);
  if (rc == 0) {
    P.issued[pick]++;
    P.pending.push_back(sm);
  } else {
    (void)hipEventDestroy(sm.e0);
    (void)hipEventDestroy(sm.e1);
  }
  return rc;
#undef NSOL_GO
}

// returns -2 if the kernel does not apply to this problem
template <typename T>
int fusedk_impl(const T *xbar_in, T *xbar_out, const T *x_in, T *x_out, const T *bt,
                const T *p_in, T *p_out, int ndim, int64_t nz, int64_t ny,
                int64_t nx, double wx, double wy, double wz, int k,
                const double *sigma, const double *hden, const double *tau,
                const double *tl, const double *theta, int flags, void *stream,
                int64_t pitch = 0) {
  NSOL_CHECK_GEOM(ndim, nz, ny, nx);
  if (!xbar_in || !xbar_out || !x_in || !x_out || !bt || !p_out || !sigma ||
      !hden || !tau || !tl || !theta || xbar_in == xbar_out || p_in == p_out ||
      x_in == x_out || k < 2 || k > 3)
    return NSOL_EINVAL;
  constexpr int VW = 16 / sizeof(T);
  if (pitch > 0 && (pitch < nx || pitch % VW != 0)) return NSOL_EINVAL;
  // (rows that are not a multiple of 16 bytes take the RAG instantiation: their
  // accesses are 4-byte aligned anyway, so only the element size matters then)
  const bool rag = nx % VW != 0;
  if (!g_tunek.enable || k > g_tunek.kmax || ndim != 3 ||
      nz * ny * nx < ((int64_t)g_tunek.min_kvox << 10) ||
      nx / VW < 8 || ny < 8 || nz < 8 || nz >= ((int64_t)1 << 30) ||
      (!rag && (!al16(xbar_in) || !al16(xbar_out) || !al16(x_in) || !al16(x_out) ||
                !al16(bt) || !al16(p_out) || (p_in && !al16(p_in)) ||
                (nz * ny * nx) % VW != 0)))
    return -2;
  const Geom<T> G = make_geom_pitched<T>(ndim, nz, ny, nx, pitch, wx, wy, wz);
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
  else if (!strcmp(name, "pdk_verbose")) nsol_pdk::g_tunek.verbose = value;
  else if (!strcmp(name, "pdk_autotune")) nsol_pdk::g_tunek.autotune = value;
  else if (!strcmp(name, "pdk_pf2")) nsol_pdk::g_tunek.pf2 = value;
  else if (!strcmp(name, "pdk_split")) nsol_pdk::g_split = value;
  else if (!strcmp(name, "pdk_min_kvox")) nsol_pdk::g_tunek.min_kvox = value;
  else if (!strcmp(name, "pdk_tail2")) nsol_pdk::g_tunek.tail2 = value;
  else if (!strcmp(name, "pdk_tune_min_mvox")) nsol_pdk::g_tunek.tune_min_mvox = value;
  else if (!strcmp(name, "pdk_forget")) {
    std::lock_guard<std::mutex> lock(nsol_pdk::g_plans_mutex);
    for (auto &kv : nsol_pdk::g_plans)
      for (auto &sm : kv.second.pending) {
        (void)hipEventDestroy(sm.e0);
        (void)hipEventDestroy(sm.e1);
      }
    nsol_pdk::g_plans.clear();
  }
  else return NSOL_EINVAL;
  return 0;
}

int nsol_pd_fusedk_tuned(int elem_size, int k, int64_t nz, int64_t ny, int64_t nx) {
  std::lock_guard<std::mutex> lock(nsol_pdk::g_plans_mutex);
  auto it = nsol_pdk::g_plans.find(
      nsol_pdk::PlanKey{nsol_pdk::current_device(), elem_size, k, nz, ny, nx});
  if (it == nsol_pdk::g_plans.end()) return -1;
  nsol_pdk::plan_poll(it->second, k);
  return it->second.chosen >= 0 ? 1 : 0;
}

/* 1 when a run's trailing pair of iterations should go through depth 2 of k_pd_fusedk
 * (the shape's depth-3 plan has settled and the knob is on), else 0 */
extern "C" int nsol_pd_fusedk_tail2_pitched(int elem_size, int64_t nz, int64_t ny, int64_t nx,
                                            int64_t pitch);
extern "C" int nsol_pd_fusedk_tail2(int elem_size, int64_t nz, int64_t ny, int64_t nx) {
  return nsol_pd_fusedk_tail2_pitched(elem_size, nz, ny, nx, 0);
}
extern "C" int nsol_pd_fusedk_tail2_pitched(int elem_size, int64_t nz, int64_t ny, int64_t nx,
                                            int64_t pitch) {
  if (!nsol_pdk::g_tunek.tail2 || !nsol_pdk::g_tunek.enable || nsol_pdk::g_tunek.kmax < 3 ||
      nz * ny * nx < ((int64_t)nsol_pdk::g_tunek.tune_min_mvox << 20))
    return 0;                       /* (measured on volumes the online tuner handles) */
  std::lock_guard<std::mutex> lock(nsol_pdk::g_plans_mutex);
  auto it = nsol_pdk::g_plans.find(
      nsol_pdk::PlanKey{nsol_pdk::current_device(), elem_size, 3, nz, ny, nx,
                        pitch > nx ? pitch : 0});
  return it != nsol_pdk::g_plans.end() && it->second.chosen >= 0 ? 1 : 0;
}

int nsol_pd_fusedk_launches(int k) {
  if (k != 2 && k != 3) return -1;
  return nsol_pdk::g_launches[k - 2].load(std::memory_order_relaxed);
}

int nsol_pd_fusedk_plan(int elem_size, int k, int64_t nz, int64_t ny, int64_t nx,
                        int *waves, int *ntx, int64_t *zchunk) {
  std::lock_guard<std::mutex> lock(nsol_pdk::g_plans_mutex);
  auto it = nsol_pdk::g_plans.find(
      nsol_pdk::PlanKey{nsol_pdk::current_device(), elem_size, k, nz, ny, nx});
  if (it == nsol_pdk::g_plans.end() || it->second.chosen < 0) return NSOL_EINVAL;
  const nsol_pdk::Config &c = it->second.cand[it->second.chosen];
  if (waves) *waves = c.nw;
  if (ntx) *ntx = c.q.ntx;
  if (zchunk) *zchunk = c.zchunk;
  return 0;
}

#define NSOL_PDK_PITCHED(T, SUF)                                                       \
  int nsol_pd_fusedk_iter_pitched_##SUF(                                               \
      const T *xbar_in, T *xbar_out, const T *x_in, T *x_out, const T *bt, const T *p_in, \
      T *p_out, int ndim, int64_t nz, int64_t ny, int64_t nx, int64_t pitch, double wx, \
      double wy, double wz, int k, const double *sigma, const double *hden,            \
      const double *tau, const double *tl, const double *theta, int flags,             \
      void *stream) {                                                                  \
    return nsol_pdk::fusedk_impl<T>(xbar_in, xbar_out, x_in, x_out, bt, p_in, p_out,   \
                                    ndim, nz, ny, nx, wx, wy, wz, k, sigma, hden, tau, \
                                    tl, theta, flags, stream, pitch);                  \
  }
NSOL_PDK_PITCHED(float, f32)
NSOL_PDK_PITCHED(double, f64)
#undef NSOL_PDK_PITCHED

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
