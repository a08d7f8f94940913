// The one-pass blur with the input staged by LDS-DMA: kernel and launcher, compiled
// once per element type (nsol_blur3_f32.hip, nsol_blur3_f64.hip) and called from
// nsol_conv.hip -- the instantiations (tap counts x isotropic x epilogue x ragged)
// are most of the library's compile time.
#pragma once
#include <type_traits>

#include "nsol_common.hpp"

// experiment knobs (defined in nsol_conv.hip, set through nsol_hip_set_param)
extern "C" __attribute__((visibility("hidden"))) int nsol_blur3_zchunk;
extern "C" __attribute__((visibility("hidden"))) int nsol_blur3_dma_rag;

namespace nsol_blur3 {

using namespace nsol;

constexpr int kMaxTaps = 129;

template <typename T>
struct Taps {
  T w[kMaxTaps];
};

template <typename T, int V>
struct VecOf {
  typedef T type __attribute__((ext_vector_type(V)));
};

// Raw buffer addressing: a 32-bit byte offset per lane plus a scalar plane offset; a
// lane whose offset is kNoLane is out of range of every buffer, so its load returns
// 0 and its store is dropped without touching memory.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr uint32_t kNoLane = 0xC0000000u;

constexpr int kDmaLxb = 16;   // lanes per tile row of the LDS-DMA staged kernel

// Cache policy of the LDS-DMA loads (the builtin's aux operand: 0 = default, 2 = nt):
// own-position tiles are read by ONE workgroup once (y_prev, q0, b, the old io); the
// raw tile's and the halo'd tile's edges are read again by the neighbouring tiles.
#ifndef NSOL_B3_AUX_OWN
#define NSOL_B3_AUX_OWN 0
#endif
#ifndef NSOL_B3_AUX_HALO
#define NSOL_B3_AUX_HALO 0
#endif
#ifndef NSOL_B3_AUX_RAW
#define NSOL_B3_AUX_RAW 0
#endif

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

// what nsol_conv.hip calls: -2 when the kernel does not apply (nothing launched).
// epi 1: io = ca * blur(x) + cb * io in place (out = io) and *result = the sum of
// squares of the new io (part: >= tiles doubles of scratch).  epi 2: out = blur(x),
// result[0] = the sum of squares of out and result[1] = ca * sum (d_x x)^2 +
// cb * sum (d_y x)^2 + cc * sum (d_z x)^2 (forward differences, zero behind the last
// voxel of an axis) of the INPUT (part: >= 2 * tiles doubles)
__attribute__((visibility("hidden")))
int blur3_dma_run(const float *x, float *out, int64_t nz, int64_t ny, int64_t nx,
                  const Taps<float> &tz, const Taps<float> &ty, const Taps<float> &tx,
                  int ntaps, int epi, double ca, double cb, double cc, double *result,
                  double *part, int64_t part_doubles, hipStream_t st);
__attribute__((visibility("hidden")))
int blur3_dma_run(const double *x, double *out, int64_t nz, int64_t ny, int64_t nx,
                  const Taps<double> &tz, const Taps<double> &ty, const Taps<double> &tx,
                  int ntaps, int epi, double ca, double cb, double cc, double *result,
                  double *part, int64_t part_doubles, hipStream_t st);

// The two halves of a Lanczos step on A'A + rho B'B (nsol_amd/lsmr.py, lsmr_normal) taken
// by the blur itself, unit spacing, B = gradient (rho_g) or identity (rho_i):
//   half A (epi 3):  t = A y with sum t^2 and sum |grad y|^2 (as epi 2) and
//                    q0 = c1 K'K y + c0 y + c2 y_prev            (y_prev may be null)
//   half B (epi 4):  y_new = ca A t + q0 + cy y with sum y_new^2
// The coefficients live in device memory (coef, element type T: [0..2] = c1, c0, c2 read
// by half A, [4..5] = ca, cy read by half B) and are written by the reduction kernels that
// close each half from the sums on the scalar board (doubles: board[3 j] = |y_j|^2,
// [3 j + 1] = |A y_j|^2, [3 j + 2] = |grad y_j|^2), so consecutive steps are enqueued
// without the host seeing a scalar.  -2: does not apply (nothing launched).
__attribute__((visibility("hidden")))
int blur3_lanczos_a(const float *y, const float *y_prev, float *t, float *q0, int64_t nz,
                    int64_t ny, int64_t nx, const Taps<float> &tz, const Taps<float> &ty,
                    const Taps<float> &tx, int ntaps, double rho_g, double rho_i,
                    double *board, int step, float *coef, double *part,
                    int64_t part_doubles, hipStream_t st);
__attribute__((visibility("hidden")))
int blur3_lanczos_a(const double *y, const double *y_prev, double *t, double *q0, int64_t nz,
                    int64_t ny, int64_t nx, const Taps<double> &tz, const Taps<double> &ty,
                    const Taps<double> &tx, int ntaps, double rho_g, double rho_i,
                    double *board, int step, double *coef, double *part,
                    int64_t part_doubles, hipStream_t st);
__attribute__((visibility("hidden")))
int blur3_lanczos_b(const float *t, const float *q0, const float *y, float *y_new,
                    int64_t nz, int64_t ny, int64_t nx, const Taps<float> &tz,
                    const Taps<float> &ty, const Taps<float> &tx, int ntaps, double rho_g,
                    double rho_i, double *board, int step, float *coef, double *part,
                    int64_t part_doubles, hipStream_t st);
__attribute__((visibility("hidden")))
int blur3_lanczos_b(const double *t, const double *q0, const double *y, double *y_new,
                    int64_t nz, int64_t ny, int64_t nx, const Taps<double> &tz,
                    const Taps<double> &ty, const Taps<double> &tx, int ntaps, double rho_g,
                    double rho_i, double *board, int step, double *coef, double *part,
                    int64_t part_doubles, hipStream_t st);
// The data term of the robust-loss objective as the blur's epilogue (epi 5): g = rho'(r^2) r
// for r = A x - b and *result = sum rho(r^2); loss one of linear / soft_l1 / huber, s2 =
// f_scale^2, gm the Huber threshold.  -2: does not apply (nothing launched).
__attribute__((visibility("hidden")))
int blur3_loss_epilogue(const float *x, const float *b, float *g, int64_t nz, int64_t ny,
                        int64_t nx, const Taps<float> &tz, const Taps<float> &ty,
                        const Taps<float> &tx, int ntaps, int loss, double s2, double gm,
                        double *result, double *part, int64_t part_doubles, hipStream_t st);
__attribute__((visibility("hidden")))
int blur3_loss_epilogue(const double *x, const double *b, double *g, int64_t nz, int64_t ny,
                        int64_t nx, const Taps<double> &tz, const Taps<double> &ty,
                        const Taps<double> &tx, int ntaps, int loss, double s2, double gm,
                        double *result, double *part, int64_t part_doubles, hipStream_t st);
// The leaner pair: half A is the EPI 2 blur (blur3_dma_run, epi 2: t = A y, its two sums in
// result[0..1]) closed by blur3_lanczos_a2_close; half B (epi 6) forms the step's
// c1 K'K y + c0 y + c2 y_prev itself: y_new = ca A t + that + cy y with sum y_new^2.
__attribute__((visibility("hidden")))
int blur3_lanczos_b2(const float *t, const float *y, const float *y_prev, float *y_new,
                     int64_t nz, int64_t ny, int64_t nx, const Taps<float> &tz,
                     const Taps<float> &ty, const Taps<float> &tx, int ntaps, double rho_g,
                     double rho_i, double *board, int step, float *coef, double *part,
                     int64_t part_doubles, hipStream_t st);
__attribute__((visibility("hidden")))
int blur3_lanczos_b2(const double *t, const double *y, const double *y_prev, double *y_new,
                     int64_t nz, int64_t ny, int64_t nx, const Taps<double> &tz,
                     const Taps<double> &ty, const Taps<double> &tx, int ntaps, double rho_g,
                     double rho_i, double *board, int step, double *coef, double *part,
                     int64_t part_doubles, hipStream_t st);
// ... as ONE launch behind the blur: t = A y (epi 2) and the kernel that sums its partials
// onto board[3 step + 1 .. 2] and forms coef[4 .. 5] (instead of the generic reduction of
// epi 2 followed by blur3_lanczos_a2_close)
__attribute__((visibility("hidden")))
int blur3_lanczos_a2(const float *y, float *t, int64_t nz, int64_t ny, int64_t nx,
                     const Taps<float> &tz, const Taps<float> &ty, const Taps<float> &tx,
                     int ntaps, double rho_g, double rho_i, double *board, int step,
                     float *coef, double *part, int64_t part_doubles, hipStream_t st);
__attribute__((visibility("hidden")))
int blur3_lanczos_a2(const double *y, double *t, int64_t nz, int64_t ny, int64_t nx,
                     const Taps<double> &tz, const Taps<double> &ty, const Taps<double> &tx,
                     int ntaps, double rho_g, double rho_i, double *board, int step,
                     double *coef, double *part, int64_t part_doubles, hipStream_t st);
__attribute__((visibility("hidden")))
int blur3_lanczos_a2_close(const double *sums2, double *board, int step, double rho_g,
                           double rho_i, float *coef, hipStream_t st);
__attribute__((visibility("hidden")))
int blur3_lanczos_a2_close(const double *sums2, double *board, int step, double rho_g,
                           double rho_i, double *coef, hipStream_t st);
__attribute__((visibility("hidden")))
int blur3_lanczos_init(double *board, float *coef, double rho_g, double rho_i, hipStream_t st);
__attribute__((visibility("hidden")))
int blur3_lanczos_init(double *board, double *coef, double rho_g, double rho_i, hipStream_t st);

}  // namespace nsol_blur3

#ifdef NSOL_BLUR3_DMA_IMPL
namespace nsol_blur3 {
namespace {

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

// s * a and fma(s, a, c) for a wave-uniform s.  Four floats are two packed operations, and
// both take the SAME (s, s) register pair: written on the halves, the pair is one value the
// compiler keeps once -- the four-wide splat became (s, s, s, s) in four scalar registers
// per tap and coefficient, which at 13 taps pushed some seventy scalars out into lanes of a
// vector register, each read back with a v_readlane (a vector issue slot) per use.
template <typename V, typename T>
__device__ __forceinline__ V smul(T s, V a) {
  if constexpr (sizeof(T) == 4 && sizeof(V) == 16) {
    typedef T P2 __attribute__((ext_vector_type(2)));
    const P2 w = P2{s, s};
    const P2 lo = w * P2{a[0], a[1]}, hi = w * P2{a[2], a[3]};
    return V{lo[0], lo[1], hi[0], hi[1]};
  } else {
    return splat<V, T>(s) * a;
  }
}
template <typename V, typename T>
__device__ __forceinline__ V sfma(T s, V a, V c) {
  if constexpr (sizeof(T) == 4 && sizeof(V) == 16) {
    typedef T P2 __attribute__((ext_vector_type(2)));
    const P2 w = P2{s, s};
    const P2 lo = __builtin_elementwise_fma(w, P2{a[0], a[1]}, P2{c[0], c[1]});
    const P2 hi = __builtin_elementwise_fma(w, P2{a[2], a[3]}, P2{c[2], c[3]});
    return V{lo[0], lo[1], hi[0], hi[1]};
  } else {
    return __builtin_elementwise_fma(splat<V, T>(s), a, c);
  }
}

// The z window's planes as two register PAIRS per lane (four floats): held as whole
// four-wide values the two packed operations of a tap were folded back into one four-wide
// operation whose splat took four scalars again -- 28 scalar registers for the z pass alone
// beside the 14 the x and y passes share.
template <typename T, typename V, bool PK>
struct Halves {
  V v;
  __device__ __forceinline__ void set(V a) { v = a; }
  __device__ __forceinline__ V get() const { return v; }
  __device__ __forceinline__ void mul(T s, const Halves &a) { v = smul(s, a.v); }
  __device__ __forceinline__ void fma(T s, const Halves &a) { v = sfma(s, a.v, v); }
};
template <typename T, typename V>
struct Halves<T, V, true> {
  typedef T P2 __attribute__((ext_vector_type(2)));
  P2 lo, hi;
  __device__ __forceinline__ void set(V a) { lo = P2{a[0], a[1]}; hi = P2{a[2], a[3]}; }
  __device__ __forceinline__ V get() const { return V{lo[0], lo[1], hi[0], hi[1]}; }
  __device__ __forceinline__ void mul(T s, const Halves &a) {
    const P2 w = P2{s, s};
    lo = w * a.lo;
    hi = w * a.hi;
  }
  __device__ __forceinline__ void fma(T s, const Halves &a) {
    const P2 w = P2{s, s};
    lo = __builtin_elementwise_fma(w, a.lo, lo);
    hi = __builtin_elementwise_fma(w, a.hi, hi);
  }
};

__device__ __forceinline__ float fma1(float a, float b, float c) {
  return __builtin_fmaf(a, b, c);
}
__device__ __forceinline__ double fma1(double a, double b, double c) {
  return __builtin_fma(a, b, c);
}


// The robust losses that cost a handful of operations (linear, soft_l1, huber), with
// the expressions of nsol_ops.hip's loss_eval: rho(f2) and rho'(f2) for f2 = r^2.
template <typename T>
__device__ __forceinline__ void blur3_loss(int loss, T f2, T s2, T gm, T &rho, T &drho) {
  const T z = f2 / s2;
  if (loss == NSOL_LOSS_SOFT_L1) {
    const T q = t_sqrt(T(1) + z);
    rho = T(2) * (q - T(1)) * s2;
    drho = T(1) / q;
  } else if (loss == NSOL_LOSS_HUBER) {
    const T g2 = gm * gm;
    if (z < g2) { rho = z * s2; drho = T(1); }
    else {
      const T q = t_sqrt(z);
      rho = (T(2) * gm * q - g2) * s2;
      drho = gm / q;
    }
  } else {
    rho = f2;
    drho = T(1);
  }
}

// LDS vectors of the one-pass blur's tiles at NW waves: three raw tiles (whole 1-KiB
// pieces) and two x-filtered tiles
template <int VEC, int NT, int NW>
constexpr size_t blur3_base_vecs() {
  constexpr int R = NT / 2, NBH = (R + VEC - 1) / VEC;
  constexpr int tyr = NW * 64 / kDmaLxb, frows = tyr + 2 * R;
  return 3 * (size_t)((frows * (kDmaLxb + 2 * NBH) + 63) / 64) * 64 +
         2 * (size_t)frows * kDmaLxb;
}
// EPI 6: a halo'd tile of y per wave (128 slots each) where that fits beside the
// own-position tile of y_prev; else one shared tile (see the kernel)
#ifndef NSOL_B3_STEADY
#define NSOL_B3_STEADY 0             // (1: whole trips of steady-state phases take an instantiation
                                     // of their own; measured with the plain blur and EPI 2 only --
                                     // EPI 6 has no registers for it: 308 bytes of scratch, whose
                                     // loads and stores would also break the counted waits)
#endif
#ifndef NSOL_B3_EPI6_MINI
#define NSOL_B3_EPI6_MINI 1          // (0: the shared tile everywhere, for A/B runs)
#endif
template <typename T, int VEC, int NT, int NW>
constexpr bool blur3_epi6_mini() {
  return NSOL_B3_EPI6_MINI && kDmaLxb == 16 &&
         (blur3_base_vecs<VEC, NT, NW>() + (size_t)NW * 64 + (size_t)NW * 128) * 16 <=
             160 * 1024;
}

// phases U .. M-1 of one trip through the loop body (each with its position in the
// ring as a compile-time constant); stops at the end of the z chunk
// (S: a trip whose phases are all in the steady state -- see the kernel's loop)
template <int U, int M, bool S, typename F>
__device__ __forceinline__ void blur3_phases(int st0, int nsteps, F &f) {
  if constexpr (U < M) {
    if (S || st0 + U < nsteps) {
      f(st0 + U, std::integral_constant<int, U>(), std::integral_constant<bool, S>());
      blur3_phases<U + 1, M, S>(st0, nsteps, f);
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
//
// EPI == 2: A x is stored as it is, with its sum of squares (part[tile]) and, from the
// raw tiles the x pass reads anyway, the weighted sum of squares of the forward
// differences of the INPUT, ca |d_x x|^2 + cb |d_y x|^2 + cc |d_z x|^2 (part[tiles +
// tile]; zero behind the last voxel of an axis, as nsol_grad_* has it, not the blur's
// periodic wrap) -- the two sums of a Lanczos step on A'A + rho grad'grad
// (nsol_amd/lsmr.py, lsmr_normal) without a second read of x.  A lane reads its own
// vector of plane st + 1, the vector to its right and the one below from the raw tile
// and keeps its own vector of the plane before in registers for d_z.
//
// EPI == 3 / 4: the two halves of a Lanczos step (see blur3_lanczos_a / _b above; unit
// spacing, no RAG form).  EPI 3 is EPI 2 plus q0 = c1 K'K x + c0 x + c2 aux1 stored to
// aux_out: the in-plane part of K'K x of plane st + 1 comes from the raw tile the
// difference sums read anyway (plus the vector to the left and the one above), the z part
// of plane st from the lane's own vectors of planes st - 1, st, st + 1 kept in registers;
// every difference and sum is formed in the order of k_tk1_reg<.., 2> (nsol_lsmr.hip), so
// q0 equals that kernel's result bit for bit.  EPI 4 stores ca A x + aux1 + cy aux2 (the
// order of nsol_lincomb3_*) with its sum of squares.  The own-position tiles of aux1 /
// aux2 travel global -> LDS by LDS-DMA, requested at the start of a phase ahead of its
// raw-tile pieces and stores, so the counted wait at the end of the phase has them landed
// while those stay in flight.  EPI 3 alternates two tiles like EPI 1; EPI 4 has LDS for
// ONE tile per array: a lane moves its values of this plane to registers first and the
// next plane's piece is requested into the same tile (a wave only ever reads what its own
// piece brought).
//
// EPI == 6: the second half of a Lanczos step WITH the step's K'K y (so that the first half
// is EPI 2 and no q0 ever goes to memory: 25 B per voxel and step instead of 33):
//   y_new = ca A t + (c1 K'K y + c0 y + c2 y_prev) + cy y,  sum y_new^2
// with aux1 = y, aux2 = y_prev (may be null).  Plane j + 1 of y travels into ONE tile WITH
// a one-vector / one-row halo while output plane j is worked on; a lane takes its own
// vector and the in-plane part of K'K y of that plane from it at the start of a phase (the
// values of EPI 3, in its order) and keeps y of the two planes before in registers for the
// z part.  The neighbours come from other waves' pieces, so a barrier separates those reads
// from the request for the next plane into the same tile (two tiles would not fit the LDS).
//
// EPI == 5: the data term of the robust-loss objective (tikhonov_linear_solver.py:
// 201-208) as the epilogue of A x: with b = aux1 staged like EPI 1's old io tile, the
// kernel stores rho'(r^2) r for r = A x - b and leaves sum rho(r^2) in part[tile] --
// what nsol_loss_cost_grad_* makes of a stored A x in a pass of its own (ca = f_scale^2,
// cb = the Huber threshold, cc = the loss; the element-wise values are that kernel's bit
// for bit, the sum is taken in another order).  No RAG form.
//
// RAG: rows that are not a multiple of 16 bytes (or operands that are not 16-byte
// aligned).  LDS-DMA takes 16-byte pieces from any 4-byte aligned source and honours
// EXEC (tools/_probe/dma_probe.hip), so the raw tile is staged as before from
// element-aligned sources, except for the ONE slot per raw row that straddles the end
// of the volume's row (elements nx-k .. nx-1 followed by 0 ..): its lane sits the
// 16-byte piece out, and the slot is filled by a 4-byte LDS-DMA piece (lanes 0 - 3 of
// one wave, one instruction per raw row) -- only in the tiles whose window holds the
// row end.  The row's last, partial output vector is stored element by element (its
// 16-byte store carries the out-of-range offset): VEC - 1 more stores per wave and
// plane in the last tile column, all of them counted by the waits.
template <typename T, int VEC, int NT, int NW, bool ISO, int EPI = 0, bool RAG = false>
__global__ __launch_bounds__(NW * 64) void k_blur3_dma(
    const T *__restrict__ x, T *__restrict__ out, int64_t nz, int64_t ny, int64_t nx,
    Taps<T> tz_, Taps<T> ty_, Taps<T> tx, int ntx, int nty, int nzc, int zchunk,
    int per_xcd, T ca = T(1), T cb = T(0), T cc = T(0),
    double *__restrict__ part = nullptr, const T *__restrict__ aux1 = nullptr,
    const T *__restrict__ aux2 = nullptr, T *__restrict__ aux_out = nullptr,
    const T *__restrict__ coef = nullptr) {
  static_assert(!(RAG && EPI >= 3), "the Lanczos halves have no ragged form");
  static_assert(EPI < 3 || NT >= 5, "the Lanczos halves keep two planes of history");
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
#ifdef NSOL_B3_PERMUTE_BX        // (measurement only: tile column != dispatch slot; 8 columns)
  const int bxs = logical % ntx;
  const int bx = ntx == 8 ? ((bxs & 1) << 2 | (bxs & 2) | (bxs & 4) >> 2) : bxs;
#else
  const int bx = logical % ntx;
#endif
  const int by = (logical / ntx) % nty;
  const int bz = logical / (ntx * nty);

  constexpr int tyr = NTHR / lxb;              // rows of the tile = rows of lanes
  constexpr int frows = tyr + 2 * R;           // rows of the raw / x-filtered tile
  constexpr int rl = lxb + 2 * NBH;            // vectors per raw row
  constexpr int raw_vecs = frows * rl;
  constexpr int npieces = (raw_vecs + 63) >> 6;  // 1 KiB per wave-instruction
  // (RAG) behind every raw tile, one 16-byte slot per raw row for the vector that
  // straddles the end of the volume's row: KPW wave-instructions of 4-byte pieces
  constexpr int KPW = RAG ? (frows * 4 + 63) / 64 : 0;
  constexpr int patch0 = npieces * 64;         // first patch slot of a raw buffer
  constexpr int raw_stride = npieces * 64 + KPW * 16;   // vectors per raw buffer
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
  static_assert(lxb == 16, "a wave covers 4 rows of 16 lanes");
  const int lx = lane & 15;
  const int quad = (lane >> 2) & 3;
  const int rsel = ((quad == 1 || quad == 2) ? 1 : 0) ^ ((lane >> 4) & 1);
  const int row = (tid >> 6) * 4 + ((lane >> 5) & 1) * 2 + rsel;
  const int nxv = (int)((nx + (RAG ? VEC - 1 : 0)) / VEC);
  const int xv = bx * lxb + lx;
  const int64_t y0 = (int64_t)by * tyr;
  const bool owner = xv < nxv && (y0 + row < ny);
  const int64_t plane = ny * nx;
  const int plane_i = (int)plane;              // (< 2^31: checked by the launcher)
  // (RAG) elements of this lane's output vector inside the row; the tile column that
  // holds the row's partial vector; the raw-row slot that straddles the row end
  const int nvalid = RAG ? (int)(nx - (int64_t)xv * VEC) : VEC;
  const bool tail_tile = RAG && (nx % VEC != 0) && bx == ntx - 1;
  int cs = -1;
  if constexpr (RAG) {
    const int64_t p0 = ((int64_t)bx * lxb - NBH) * VEC;
    if (nx % VEC != 0 && nx > p0 && nx < p0 + (int64_t)(lxb + 2 * NBH) * VEC)
      cs = (int)((nx - p0) / VEC);
  }

  // LDS-DMA source offsets (elements inside a plane) of this lane's pieces:
  // piece k = wave + j * NW covers the raw vectors [64 k, 64 k + 64)
  uint32_t src_off[MAXP];
  uint32_t strad = 0;                          // (RAG) pieces this lane sits out
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
    if constexpr (RAG) {
      int64_t pp = ((int64_t)bx * lxb - NBH + cc) * VEC % nx;
      if (pp < 0) pp += nx;
      if (pp + VEC > nx) { strad |= 1u << j; pp = 0; }
      src_off[j] = (uint32_t)(yy * nx + pp);
    } else {
      int xx = (bx * lxb - NBH + cc) % nxv;
      if (xx < 0) xx += nxv;
      src_off[j] = (uint32_t)(yy * nx + (int64_t)xx * VEC);
    }
  }
  // (RAG) the 4-byte pieces of the straddling slots: wave w < KPW fills the patch
  // slots of the raw rows 16 w .. 16 w + 15, lane l dword l % 4 of row 16 w + l / 4
  constexpr int DW = (int)(sizeof(T) / 4);
  uint32_t pat_off = 0, pat_sub = 0;
  int npat = 0;
  if constexpr (RAG) {
    if (cs >= 0 && wave < KPW) {
      npat = 1;
      const int64_t xs = ((int64_t)bx * lxb - NBH + cs) * VEC;   // < nx < xs + VEC
      int64_t xe = xs + (lane & 3) / DW;
      if (xe >= nx) xe -= nx;
      pat_sub = (uint32_t)((lane & 3) % DW);
      int rr = wave * 16 + (lane >> 2);
      if (rr >= frows) rr = frows - 1;               // (lands in the padding)
      int64_t yy = (y0 - R + rr) % ny;
      if (yy < 0) yy += ny;
      pat_off = (uint32_t)(yy * nx + xe);
    }
  }
  // pieces this wave issues per plane (wave-uniform)
  const int my_pieces = (npieces - wave + NW - 1) / NW;
  // (plane indices are 32-bit: a 64-bit comparison of scalars is a vector instruction)
  auto stage = [&](int z, int rbuf) {          // plane z -> raw tile at vector offset rbuf
    const T *pl = x + (int64_t)z * plane_i;
#pragma unroll
    for (int j = 0; j < MAXP; ++j) {
      if (j * NW >= npieces) break;              // (compile time)
      const int k = wave + j * NW;
      if ((j + 1) * NW <= npieces || k < npieces) {
        if (!RAG || !((strad >> j) & 1u))
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void *)(pl + src_off[j]),
              (__attribute__((address_space(3))) void *)(raw + (size_t)rbuf + (size_t)k * 64),
              16, 0, NSOL_B3_AUX_RAW);
      }
    }
    if constexpr (RAG) {
      if (npat)                                    // (wave-uniform)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void *)(
                reinterpret_cast<const uint32_t *>(pl + pat_off) + pat_sub),
            (__attribute__((address_space(3))) void *)(raw + (size_t)rbuf + patch0 +
                                                       (size_t)wave * 16),
            4, 0, 0);
    }
  };
  // vector-memory operations one staged plane / one stored plane costs this wave
  const int my_stage_ops = my_pieces + npat;
  const int nv_tail = RAG ? (int)(nx % VEC) : 0;   // elements of the row's partial vector
  const int my_store_ops =
      1 + (tail_tile ? (VEC == 4 ? ((nv_tail >> 1) & 1) + (nv_tail & 1) : 1) : 0);
  // End of a phase: the LDS writes of this phase are done (lgkmcnt) and at most
  // `newer` of this wave's vector-memory operations are still in flight.  On gfx9
  // vmcnt is decremented in issue order for loads and stores alike, so with
  // `newer` = the number of operations issued after the pieces of the tile that has
  // to be complete (<= MAXP pieces + 1 store), that tile has landed.  Then the barrier.
  auto phase_end = [&](int newer) {                // (wave-uniform)
#define NSOL_B3_WAIT(N) \
  asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" ::: "memory")
    switch (newer) {
      case 0: NSOL_B3_WAIT(0); break;
      case 1: NSOL_B3_WAIT(1); break;
      case 2: NSOL_B3_WAIT(2); break;
      case 3: NSOL_B3_WAIT(3); break;
      case 4: NSOL_B3_WAIT(4); break;
      default:
        if constexpr (!RAG && EPI != 6) { NSOL_B3_WAIT(4); break; }
        switch (newer) {                  // (a smaller count only waits longer)
          case 5: NSOL_B3_WAIT(5); break;
          case 6: NSOL_B3_WAIT(6); break;
          case 7: NSOL_B3_WAIT(7); break;
          case 8: NSOL_B3_WAIT(8); break;
          case 9: NSOL_B3_WAIT(9); break;
          case 10: NSOL_B3_WAIT(10); break;
          case 11: NSOL_B3_WAIT(11); break;
          default: NSOL_B3_WAIT(12); break;
        }
    }
#undef NSOL_B3_WAIT
  };
  // x pass: raw tile -> x-filtered tile; a lane filters footprint row `row` and,
  // in the first waves, the halo row `tyr + row`
  const bool second = row < 2 * R;
  const bool second_wave = __builtin_amdgcn_readfirstlane((int)second) != 0;
  // (PS: a tile whose raw rows hold the straddling slot cs reads that one vector from
  // the row's patch slot)
  auto xrow = [&](const V *rb, V *xb, int fr, auto PS) {
    constexpr bool ps = decltype(PS)::value;
    const V *w = rb + (size_t)fr * rl + lx;
    const V *pslot = rb + patch0 + fr;
    auto wload = [&](int b) -> V {
      if constexpr (ps) return *((lx + b == cs) ? pslot : w + b);
#ifdef NSOL_B3_ABLATE_XREADS       // (timing only: 3 vectors read for the window's 5)
      else return w[b & ~1];
#else
      else return w[b];
#endif
    };
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
        const V t = wload(b);
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
        const V t = wload(b);
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
    if (RAG && cs >= 0) {                          // (uniform in the workgroup)
      xrow(rb, xb, row, std::true_type());
      if (second_wave) {
        if (second) xrow(rb, xb, row + tyr, std::true_type());
      }
    } else {
      xrow(rb, xb, row, std::false_type());
      if (second_wave) {
        if (second) xrow(rb, xb, row + tyr, std::false_type());
      }
    }
  };

  const int nzi = (int)nz;                          // (< 2^31: checked by the launcher)
  const int zbeg = bz * zchunk;
  int zend = zbeg + zchunk;
  if (zend > nzi) zend = nzi;
  const int nsteps = (zend - zbeg) + 2 * R;    // planes zbeg - R .. zend + R - 1
  int zw = (zbeg - R) % nzi;                        // plane of the next stage()
  if (zw < 0) zw += nzi;
  auto next_plane = [&]() {
    const int z = zw;
    if (++zw == nzi) zw = 0;
    return z;
  };
  typedef Halves<T, V, sizeof(T) == 4 && VEC == 4> RV;
  RV ring[NT - 1];                                  // xy-filtered planes, oldest first
#pragma unroll
  for (int t = 0; t + 1 < NT; ++t) ring[t].set(splat<V, T>(T(0)));
  // output: one buffer descriptor per plane and a 32-bit offset inside it; lanes
  // that own no voxel carry an out-of-range offset (the store is dropped by the
  // hardware), so EVERY wave issues exactly one store per output plane -- the
  // counted waits below depend on that
  const uint32_t plane_bytes = (uint32_t)(plane * sizeof(T));
  const uint32_t vec_off = (uint32_t)(((y0 + row) * nx + (int64_t)xv * VEC) * sizeof(T));
  const uint32_t own_off = owner && nvalid >= VEC ? vec_off : kNoLane;
  const bool tail = RAG && owner && nvalid < VEC;
  const int rot = RAG ? (int)(VEC - nx % VEC) % VEC : 0;   // (EPI) see stage_old
  double sumsq = 0.0;
  // (EPI == 2) weights of the squared differences, zero in lanes that own no voxel;
  // 0 / 1 factors that blank the neighbour behind the volume's last column and row
  // (both sums of this form gather in T over the planes of one trip through the loop
  // body -- 4 and 12 squares per plane and lane -- and are widened once per trip:
  // per plane, six conversions and six double operations were a tenth of the
  // kernel's issue slots)
  double gsum = 0.0;
  T sacc = T(0), gacc = T(0);
  V prev_own = splat<V, T>(T(0));
  const T gx2 = owner ? ca : T(0), gy2 = owner ? cb : T(0), gz2 = owner ? cc : T(0);
  const T xm = (xv + 1 < nxv) ? T(1) : T(0);
  const T ym = (y0 + row + 1 < ny) ? T(1) : T(0);
  // (EPI 3) the lane's own vectors of the two planes before, the in-plane part of K'K x
  // of the plane before, 0 / 1 factors for the backward differences at the first column
  // and row; (EPI 3 / 4) the coefficients, read once
  V prev2_own = splat<V, T>(T(0)), lapxy_prev = splat<V, T>(T(0));
  const T lm = xv > 0 ? T(1) : T(0);
  const T um = (y0 + row > 0) ? T(1) : T(0);
  T k0 = T(0), k1 = T(0), k2 = T(0);
  if constexpr (EPI == 3) { k0 = coef[0]; k1 = coef[1]; k2 = coef[2]; }   // c1, c0, c2
  if constexpr (EPI == 4) { k0 = coef[4]; k1 = coef[5]; }                   // ca, cy
  T k3 = T(0), k4 = T(0);
  if constexpr (EPI == 6) {                                                 // c1, c0, c2, ca, cy
    k0 = coef[0]; k1 = coef[1]; k2 = coef[2]; k3 = coef[4]; k4 = coef[5];
  }
  const bool has_prev = (EPI == 3 && aux1 != nullptr) || (EPI == 6 && aux2 != nullptr);
  // (EPI 6) y with a one-vector / one-row halo, behind the own-position tile.  Where the
  // LDS has room (MINI) every wave keeps a halo'd tile of ITS OWN four rows -- 6 rows of
  // lxb + 2 vectors in two 1-KiB pieces (the second one's last 20 lanes land in padding)
  // -- and reads nothing another wave's piece brought: no barrier between those reads and
  // the request for the next plane, as with EPI 4's own-position tiles.  Otherwise ONE
  // tile of (tyr + 2) rows is shared (piece k = wave + j * NW of it, two at most) and a
  // barrier separates its reads from the next request.  Sources are clamped into the plane.
  constexpr int hrl = lxb + 2;
  constexpr bool MINI = blur3_epi6_mini<T, VEC, NT, NW>();
  static_assert(!MINI || (lxb == 16 && 6 * hrl <= 128), "a wave's halo'd rows: two pieces");
  constexpr int halo_vecs = MINI ? 6 * hrl : (tyr + 2) * hrl;
  constexpr int hpieces = (halo_vecs + 63) >> 6;
  static_assert(EPI != 6 || MINI || hpieces <= 2 * NW, "halo tile: two pieces per wave");
  uint32_t yh_off[2] = {0, 0};
  if constexpr (EPI == 6) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int i = MINI ? j * 64 + lane : (wave + j * NW) * 64 + lane;
      if (i >= halo_vecs) i = 0;
      // (Positions outside the volume are WRAPPED like the raw tile's, not clamped: what
      // they hold is never used -- the edge masks select the boundary's form -- but a
      // clamped position repeats a line the tile holds anyway, and the tile columns at
      // the volume's edges then ask for a tenth fewer lines than the others, run 9 %
      // ahead of them (7 phases after 70) and the halo lines neighbouring columns share
      // stop meeting in the L2: tools/_probe/drift_probe.py, HISTORY.)
      int64_t yy = y0 + (MINI ? wave * 4 : 0) + i / hrl - 1;
      int xx = bx * lxb + i % hrl - 1;
#ifdef NSOL_B3_CLAMP_YH           // (the clamped form, for A/B runs)
      yy = yy < 0 ? 0 : (yy >= ny ? ny - 1 : yy);
      xx = xx < 0 ? 0 : (xx >= nxv ? nxv - 1 : xx);
#else
      yy = ((yy % ny) + ny) % ny;
      xx = ((xx % nxv) + nxv) % nxv;
#endif
      yh_off[j] = (uint32_t)(yy * nx + (int64_t)xx * VEC);
    }
  }
  V ym1 = splat<V, T>(T(0)), y0c = splat<V, T>(T(0)), lap0 = splat<V, T>(T(0));
  auto put = [&](int z, V val, int ob) {
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out + (int64_t)z * plane_i, 0, plane_bytes,
                                                        0x00020000);
    if constexpr (EPI == 1) {
      V old = obuf[(size_t)ob * tile_vecs + (size_t)row * lxb + lx];
      if constexpr (RAG) {
        if (tail) {                                  // staged from nx - VEC: old[e] = L[e + rot]
#pragma unroll
          for (int k = 0; k < VEC - 1; ++k)
            if (k < rot) {
              const T f = old[0];
#pragma unroll
              for (int e = 0; e + 1 < VEC; ++e) old[e] = old[e + 1];
              old[VEC - 1] = f;
            }
        }
      }
      val = smul(ca, val) + smul(cb, old);
    }
    if constexpr (EPI == 5) {
      const V bv = obuf[(size_t)ob * tile_vecs + (size_t)row * lxb + lx];
      const int kind = (int)cc;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const T r = T(1) * val[e] + T(-1) * bv[e];          // (as nsol_lincomb2 forms A x - b)
        T rho, drho;
        blur3_loss(kind, r * r, ca, cb, rho, drho);
        if (owner) sumsq += (double)rho;
        val[e] = drho * r;
      }
    }
    if constexpr (EPI != 0 && EPI != 5) {
      if (owner) {
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if (!RAG || e < nvalid) {
            if constexpr (EPI >= 2) sacc = fma1(val[e], val[e], sacc);
            else sumsq += (double)val[e] * (double)val[e];
          }
      }
    }
    // (non-temporal: the output is not read again by this launch; 0.2285 -> 0.2245 ms)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), rs, own_off, 0, 2);
    if constexpr (RAG) {
      if (tail_tile) {                               // (uniform in the workgroup)
        // the partial vector of the row: its first two elements as one 8-byte store,
        // the odd one as a 4-byte store (float); its one element (double)
        const uint32_t eo = tail ? vec_off : kNoLane;
        if constexpr (VEC == 4) {
          typedef T P2 __attribute__((ext_vector_type(2)));
          if (nv_tail & 2) {
            const P2 v2 = P2{val[0], val[1]};
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v2), rs, eo, 0, 0);
          }
          if (nv_tail & 1) {
            const T ve = (nv_tail & 2) ? val[2] : val[0];
            __builtin_amdgcn_raw_buffer_store_b32(
                __builtin_bit_cast(uint32_t, ve), rs,
                tail ? vec_off + (uint32_t)((nv_tail & 2) * sizeof(T)) : kNoLane, 0, 0);
          }
        } else {
          const T ve = val[0];   // (a bit_cast of the element reference itself reads element 0)
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ve), rs, eo, 0, 0);
        }
      }
    }
    asm volatile("s_nop 1");   // see nsol_pdk.hip: store data vs. the next VALU write
  };
  // (EPI) the io tile of one output plane -> obuf[ob]: one 1-KiB piece per wave; lanes
  // whose tile position lies outside the volume re-read a valid neighbour
  uint32_t old_off = 0;
  if constexpr (EPI == 1 || EPI >= 3) {
    const int i = wave * 64 + lane;
    int64_t yy = y0 + i / lxb;
    if (yy >= ny) yy = ny - 1;
    int xx = bx * lxb + i % lxb;
    if (xx >= nxv) xx = nxv - 1;
    int64_t xe = (int64_t)xx * VEC;
    // (RAG) the row's partial vector is staged from the last whole 16 bytes of the
    // row -- nothing is read past the end of io -- and rotated into place in put()
    if (RAG && xe + VEC > nx) xe = nx - VEC;
    old_off = (uint32_t)(yy * nx + xe);
  }
  auto stage_old = [&](int z, int ob) {
    if constexpr (EPI == 1 || EPI == 5)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void *)((EPI == 5 ? aux1 : out) +
                                                           (int64_t)z * plane_i +
                                                           old_off),
          (__attribute__((address_space(3))) void *)(obuf + (size_t)ob * tile_vecs +
                                                     (size_t)wave * 64),
          16, 0, NSOL_B3_AUX_OWN);
  };
  // (EPI 3 / 4) the own-position tile of plane z of `src` -> obuf[ob]
  auto stage_tile = [&](const T *src, int z, int ob) {
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void *)(src + (int64_t)z * plane_i + old_off),
        (__attribute__((address_space(3))) void *)(obuf + (size_t)ob * tile_vecs +
                                                   (size_t)wave * 64),
        16, 0, NSOL_B3_AUX_OWN);
  };
  // (EPI 6) plane z of y (clamped into the volume) with its halo -> behind the own tile
  auto stage_yh = [&](int z) {
    if (z < 0) z = 0;
    if (z >= nzi) z = nzi - 1;
    const T *pl = aux1 + (int64_t)z * plane_i;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = MINI ? wave * 2 + j : wave + j * NW;
      if (MINI || k < hpieces)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void *)(pl + yh_off[j]),
            (__attribute__((address_space(3))) void *)(obuf + (size_t)tile_vecs + (size_t)k * 64),
            16, 0, NSOL_B3_AUX_HALO);
    }
  };
  // vector-memory operations stage_yh costs this wave (wave-uniform)
  const int my_yh_ops = MINI ? 2 : (wave < hpieces ? 1 : 0) + (wave + NW < hpieces ? 1 : 0);
  (void)my_yh_ops;
  // a 16-byte store every wave issues (offset kNoLane: dropped by the hardware)
  auto store_at = [&](T *dst, int z, uint32_t off, V val) {
    if (z < 0) z = 0;
    if (z >= nzi) z = nzi - 1;
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dst + (int64_t)z * plane_i, 0, plane_bytes,
                                                        0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), rs, off, 0, 2);
    asm volatile("s_nop 1");
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
  // SS: a phase of the steady state, 2R <= st and st + max(R + 1, 3) < nsteps -- every
  // pipeline stage is at work (a plane to request, one to filter along x, one to store, the
  // difference sums and K'K inside the chunk's own planes), so none of the conditions
  // below is evaluated and the zeros of the idle cases are not merged in: the loop runs
  // its whole trips of steady phases through this instantiation (12 phases at 13 taps,
  // 10 of a 128-plane chunk's 12 trips) and keeps the general one for the first and
  // last trips.
  auto phase = [&](int st, auto U, auto S) {
    constexpr int u = decltype(U)::value;           // = st mod M
    constexpr int q = u & 1;                        // = st & 1 (M is even)
    constexpr bool SS = decltype(S)::value;
    const bool more = SS || st + 3 < nsteps;
    const bool storing = SS || st >= 2 * R;
    const bool next_x = SS || st + 1 < nsteps;      // a plane st + 1 to filter along x
    // y pass of plane st from the x-filtered tile, z pass over the ring: the value of
    // output plane st - 2R (meaningful from st = 2R on); the ring takes plane st
    auto yz_value = [&]() -> V {
      const V *col = xf + (size_t)q * xf_stride + (size_t)row * lxb + lx;
      V v = smul(ty.w[0], col[0]);
#pragma unroll
      for (int t = 1; t < NT; ++t)
#ifdef NSOL_B3_ABLATE_YREADS       // (timing only: 7 rows read for the 13 taps)
        v = sfma(ty.w[sym(t)], col[(size_t)(t & ~1) * lxb], v);
#else
        v = sfma(ty.w[sym(t)], col[(size_t)t * lxb], v);
#endif
      RV vh, acc;
      vh.set(v);
      acc.set(splat<V, T>(T(0)));
      if (storing) {
        // window of plane st: the ring from its oldest slot (u), then v
        acc.mul(tz.w[0], ring[u]);
#pragma unroll
        for (int t = 1; t < M; ++t)
#ifdef NSOL_B3_ABLATE_ZFMA         // (timing only: half of the z pass's multiply-adds)
          if (t & 1)
#endif
          acc.fma(tz.w[sym(t)], ring[(u + t) % M]);
        acc.fma(tz.w[0], vh);
      }
      ring[u] = vh;                                 // replaces plane st - M
      return acc.get();
    };
    if constexpr (EPI == 4) {
      // ---- second half of a Lanczos step: y_new = ca A x + q0 + cy y.  The lane's
      //      q0 and y of this output plane leave their (single) LDS tiles for registers
      //      at the START of the phase, and the tiles of the next output plane are
      //      requested right behind: a whole phase for them to land, like the raw tiles.
      V q0v = splat<V, T>(T(0)), yv = splat<V, T>(T(0));
      if (storing) {                                // (uniform)
        q0v = obuf[(size_t)row * lxb + lx];
        yv = obuf[(size_t)tile_vecs + (size_t)row * lxb + lx];
      }
      if (SS || (st + 1 >= 2 * R && st + 1 < nsteps)) {
        // (a wave only reads what its own piece brought: its reads above are done
        // before its next piece can land, and the wait makes that explicit)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stage_tile(aux1, zbeg + (st + 1 - 2 * R), 0);
        stage_tile(aux2, zbeg + (st + 1 - 2 * R), 1);
      }
      if (more) stage(next_plane(), r_cur);         // plane st + 3
      if (next_x) xpass(r_next, q ^ 1);
      V val = yz_value();
      uint32_t soff = kNoLane;
      if (storing) {
        val = (smul(k0, val) + q0v) + smul(k1, yv);
        if (owner) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) sacc = fma1(val[e], val[e], sacc);
        }
        soff = own_off;
      }
      store_at(out, zbeg + (st - 2 * R), soff, val);
      const int t_ = r_cur; r_cur = r_next; r_next = r_after; r_after = t_;
      phase_end((more ? my_stage_ops : 0) + 1);
      return;
    }
#ifdef NSOL_B3_DRIFT_PROBE        // (measurement only: when does each workgroup reach phases 0 / N?)
    if constexpr (EPI == 6) {
      if (tid == 0 && (st == 0 || st == NSOL_B3_DRIFT_PROBE))
        part[(st == 0 ? 3 : 2) * (size_t)total + logical] = (double)wall_clock64();
    }
#endif
    if constexpr (EPI == 6) {
      // ---- second half with K'K y: output plane j = st - 2R; the halo'd tile holds
      //      plane j + 1 of y (requested in the phase before, from phase 2R - 3 on)
      const int j = st - 2 * R;
      // (Nothing here is conditional on the phase: before the first plane has landed --
      // phases up to 2R - 3 -- the tile reads return whatever the LDS holds, and what is
      // made of it has left ym1 / y0c / lap0 again by the first phase that stores; a
      // merge of "not yet" zeros per value and phase cost more than the arithmetic.)
      // (MINI: this wave's tile, its rows 4 wave - 1 .. 4 wave + 4)
      const V *o = obuf + (size_t)tile_vecs +
                   (MINI ? (size_t)wave * 128 + (size_t)(row - wave * 4 + 1) * hrl
                         : (size_t)(row + 1) * hrl) + (lx + 1);
      const V yp1 = o[0];
      V lap1 = splat<V, T>(T(0)), ypv = splat<V, T>(T(0));
      if (k0 != T(0)) {                             // in-plane part of K'K y, plane j + 1
                                                    // (k0 = 0: B = identity, no K'K term)
        // (EPI 3's values.  Its 0 / 1 factors for the volume's edges are lane masks here,
        // which live in scalar registers -- four more vector registers and the kernel
        // spills: a * 1 - b = a - b, a * 0 - b = -b for finite a.  Both candidates are
        // formed and one is selected: left to choose, the compiler branched on the masks
        // with the tile reads inside the branches.)
        const bool e_r = xv + 1 < nxv, e_d = y0 + row + 1 < ny, e_l = xv > 0,
                   e_u = y0 + row > 0;
        const V down = o[hrl], up = o[-hrl];
        const T right0 = reinterpret_cast<const T *>(o + 1)[0],
                left3 = reinterpret_cast<const T *>(o)[-1];
        const V dyf = down - yp1, dpyf = yp1 - up;
        const T dxr = right0 - yp1[VEC - 1], dlf = yp1[0] - left3;
        T dx[VEC], dy[VEC], dpy[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          dy[k] = e_d ? dyf[k] : -yp1[k];
          dpy[k] = e_u ? dpyf[k] : T(0);
          dx[k] = (k + 1 < VEC) ? yp1[(k + 1) % VEC] - yp1[k] : (e_r ? dxr : -yp1[k]);
        }
        const T dl = e_l ? dlf : T(0);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const T l = (k > 0) ? dx[(k + VEC - 1) % VEC] : dl;
          lap1[k] = (l - dx[k]) + (dpy[k] - dy[k]);
        }
      }
      if (has_prev) ypv = obuf[(size_t)row * lxb + lx];      // (uniform)
      // every wave has taken what it needs from the two tiles: only then may the next
      // plane be requested into them (MINI: a wave reads its own pieces only -- its own
      // reads have to be done, nobody else's)
      if constexpr (MINI) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (SS || (st >= 2 * R - 3 && st + 1 < nsteps)) stage_yh(zbeg + (j + 2));
      if (has_prev && (SS || (st + 1 >= 2 * R && st + 1 < nsteps)))
        stage_tile(aux2, zbeg + (j + 1), 0);
      if (more) stage(next_plane(), r_cur);         // plane st + 3
      // q0 of the output plane now, the window moved on: four vectors fewer are live
      // through the passes below (the kernel sits at the 128 registers of 16 waves)
      V q0v;
      {
        const int zc = zbeg + j;
        const T zm = (zc + 1 < nzi) ? T(1) : T(0);
        const T zlm = zc > 0 ? T(1) : T(0);
        const V dz = sfma(zm, yp1, -y0c);
        const V dpz = smul(zlm, y0c - ym1);
        const V lap = lap0 + (dpz - dz);
        q0v = smul(k0, lap) + smul(k1, y0c);
        if (has_prev) q0v = q0v + smul(k2, ypv);    // (uniform)
      }
      ym1 = y0c; y0c = yp1; lap0 = lap1;            // (ym1: y of plane j from here on)
      if (next_x) xpass(r_next, q ^ 1);
      V val = yz_value();
      val = (smul(k3, val) + q0v) + smul(k4, ym1);
      uint32_t soff = kNoLane;
      if (storing) {
        if (owner) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) sacc = fma1(val[e], val[e], sacc);
        }
        // (aux_out set: only the sum of squares is wanted -- the last step of a solve,
        // whose vector nobody reads; the store is issued with every lane out of range)
        soff = aux_out ? kNoLane : own_off;
      }
      store_at(out, zbeg + j, soff, val);
      const int t_ = r_cur; r_cur = r_next; r_next = r_after; r_after = t_;
#ifdef NSOL_B3_ABLATE_OWNWAIT      // (timing only: the y / y_prev pieces are not waited for)
      phase_end((more ? my_stage_ops : 0) + 1 + NSOL_B3_ABLATE_OWNWAIT);
#else
      phase_end((more ? my_stage_ops : 0) + 1);
#endif
      return;
    }
    if constexpr (EPI == 3) {
      // ---- first half: K'K x of plane st (z part now, in-plane part from the phase
      //      before) -> q0; the difference sums of EPI 2 from the same values.  The
      //      y_prev tile of plane st + 1 is requested first (two tiles alternate).
      if (has_prev && (SS || (st + 1 >= R && st + 1 < nsteps - R)))
        stage_tile(aux1, zbeg + (st + 1 - R), q ^ 1);
      if (more) stage(next_plane(), r_cur);         // plane st + 3
      V q0v = splat<V, T>(T(0));
      uint32_t qoff = kNoLane;
      if (next_x) {
        const V *o = raw + (size_t)r_next + (size_t)(row + R) * rl + lx + NBH;
        const V own = o[0];
        V lapxy = splat<V, T>(T(0));
        if (SS || (st + 1 >= R && st + 1 < nsteps - R)) {  // in-plane part, plane st + 1
          const V right = o[1], left = o[-1], down = o[rl], up = o[-rl];
          const V dy = __builtin_elementwise_fma(down, splat<V, T>(ym), -own);
          const V dpy = (own - up) * splat<V, T>(um);
          T dx[VEC];
          T sx = T(0), sy = T(0);
#pragma unroll
          for (int k = 0; k < VEC; ++k) {
            dx[k] = (k + 1 < VEC) ? own[(k + 1) % VEC] - own[k]
                                  : fma1(right[0], xm, -own[k]);
            sx = fma1(dx[k], dx[k], sx);
            sy = fma1(dy[k], dy[k], sy);
          }
          gacc = fma1(gx2, sx, fma1(gy2, sy, gacc));
          const T dl = (own[0] - left[VEC - 1]) * lm;
#pragma unroll
          for (int k = 0; k < VEC; ++k) {
            const T l = (k > 0) ? dx[(k + VEC - 1) % VEC] : dl;
            lapxy[k] = (l - dx[k]) + (dpy[k] - dy[k]);
          }
        }
        if (SS || (st >= R && st < nsteps - R)) {          // z part, plane st
          const int zc = zbeg + (st - R);
          const T zm = (zc + 1 < nzi) ? T(1) : T(0);
          const T zlm = zc > 0 ? T(1) : T(0);
          const V dz = sfma(zm, own, -prev_own);
          const V dpz = smul(zlm, prev_own - prev2_own);
          T sz = T(0);
#pragma unroll
          for (int k = 0; k < VEC; ++k) sz = fma1(dz[k], dz[k], sz);
          gacc = fma1(gz2, sz, gacc);
          const V lap = lapxy_prev + (dpz - dz);
          q0v = smul(k0, lap) + smul(k1, prev_own);
          if (has_prev)                                      // (uniform)
            q0v = q0v + smul(k2, obuf[(size_t)q * tile_vecs + (size_t)row * lxb + lx]);
          qoff = own_off;
        }
        prev2_own = prev_own;
        prev_own = own;
        lapxy_prev = lapxy;
      }
      store_at(aux_out, zbeg + (st - R), qoff, q0v);
      if (next_x) xpass(r_next, q ^ 1);
      const V acc = yz_value();
      if (storing) put(zbeg + (st - 2 * R), acc, q);
      const int t_ = r_cur; r_cur = r_next; r_next = r_after; r_after = t_;
      phase_end((more ? my_stage_ops : 0) + 1 + (storing ? my_store_ops : 0));
      return;
    }
    if ((EPI == 1 || EPI == 5) && (SS || (st + 1 >= 2 * R && st + 1 < nsteps)))
      stage_old(zbeg + (st + 1 - 2 * R), q ^ 1);    // old io (b) of the next output plane
    if (more) stage(next_plane(), r_cur);           // plane st + 3, two phases ahead
    // (The x pass of plane st + 1 and the y / z passes of plane st are independent,
    // but letting half of the waves of a SIMD run them in the opposite order, so that
    // not everybody waits for the LDS at the same time, measured no faster.)
    if (next_x) xpass(r_next, q ^ 1);
    if constexpr (EPI == 2) {
      if (next_x) {
        // plane j of the chunk's nsteps planes is one of its own for R <= j < nsteps - R
        const V *rbase = raw + (size_t)r_next;
        const V *o = rbase + (size_t)(row + R) * rl + lx + NBH;
        // (RAG: the vector that straddles the row end comes from its row's patch slot,
        // and the elements of the row's partial vector that lie behind the row read as
        // zero -- every difference they enter is then the boundary's or zero)
        auto rd = [&](int dslot, int drow) -> V {
          if constexpr (RAG) {
            const V *pslot = rbase + patch0 + (row + R + drow);
            V v = *((cs >= 0 && lx + NBH + dslot == cs) ? pslot : o + drow * rl + dslot);
            if (dslot == 0) {
#pragma unroll
              for (int k = 0; k < VEC; ++k) v[k] = k < nvalid ? v[k] : T(0);
            }
            return v;
          } else {
            return o[drow * rl + dslot];
          }
        };
        const V own = rd(0, 0);
        if (SS || (st + 1 >= R && st + 1 < nsteps - R)) {  // d_x, d_y of plane st + 1
          const V right = rd(1, 0);
          const V down = rd(0, 1);
          const V dy = __builtin_elementwise_fma(down, splat<V, T>(ym), -own);
          T sx = T(0), sy = T(0);
#pragma unroll
          for (int k = 0; k < VEC; ++k) {
            const T dx = (k + 1 < VEC) ? own[(k + 1) % VEC] - own[k]
                                       : fma1(right[0], xm, -own[k]);
            sx = fma1(dx, dx, sx);
            sy = fma1(dy[k], dy[k], sy);
          }
          gacc = fma1(gx2, sx, fma1(gy2, sy, gacc));
        }
        if (SS || (st >= R && st < nsteps - R)) {          // d_z of plane st
          const T zm = (zbeg + (st - R) + 1 < nzi) ? T(1) : T(0);
          const V dz = sfma(zm, own, -prev_own);
          T sz = T(0);
#pragma unroll
          for (int k = 0; k < VEC; ++k) sz = fma1(dz[k], dz[k], sz);
          gacc = fma1(gz2, sz, gacc);
        }
        prev_own = own;
      }
    }
    {
      const V acc = yz_value();
      if (storing) put(zbeg + (st - 2 * R), acc, q);
    }
    const int t_ = r_cur; r_cur = r_next; r_next = r_after; r_after = t_;
    // plane st + 2 (staged in the previous phase) must have landed; younger than
    // its pieces are this phase's pieces and this phase's store
    phase_end((more ? my_stage_ops : 0) + (storing ? my_store_ops : 0));
  };
  const int steady_end = nsteps - (R + 1 > 3 ? R + 1 : 3);   // steady: 2R <= st < steady_end
#pragma unroll 1
  for (int st0 = 0; st0 < nsteps; st0 += M) {
    if (NSOL_B3_STEADY && EPI < 3 && st0 >= 2 * R && st0 + M - 1 < steady_end)
      blur3_phases<0, M, true>(st0, nsteps, phase);
    else
      blur3_phases<0, M, false>(st0, nsteps, phase);
    if constexpr (EPI >= 2 && EPI != 5) {
      sumsq += (double)sacc;
      gsum += (double)gacc;
      sacc = gacc = T(0);
    }
  }
  if constexpr (EPI != 0) {
    // (the last phase ended with a barrier: the LDS is free)
    double *red = reinterpret_cast<double *>(smem_raw);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sumsq += __shfl_down(sumsq, o, 64);
    if (lane == 0) red[wave] = sumsq;
    if constexpr (EPI == 2 || EPI == 3) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) gsum += __shfl_down(gsum, o, 64);
      if (lane == 0) red[NW + wave] = gsum;
    }
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int w2 = 0; w2 < NW; ++w2) t += red[w2];
      part[logical] = t;
    }
    if ((EPI == 2 || EPI == 3) && tid == 64) {
      double t = 0.0;
      for (int w2 = 0; w2 < NW; ++w2) t += red[NW + w2];
      part[total + logical] = t;
    }
  }
}

// The sum of one value per thread of a kBlock workgroup in a FIXED order (the same for
// every kernel that closes a blur: their sums are compared for equality): a shuffle tree
// inside each wave, then the waves' sums in turn; valid in thread 0.  (One thread adding
// kBlock values from the LDS one after the other took 8 of such a kernel's 12 microseconds
// -- 200 of them in a config-4 run.)
__device__ __forceinline__ double blur3_block_sum(double t, double *s) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = t;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) r += s[w];
  }
  return r;
}

// sum of the per-tile partials in a fixed order (block b: the b-th run of n partials)
__global__ __launch_bounds__(kBlock) void k_blur3_epi_final(const double *part, int n,
                                                            double *result,
                                                            double scale = 1.0) {
  __shared__ double s[kBlock / 64];
  part += (size_t)blockIdx.x * n;
  double t = 0.0;
  for (int i = threadIdx.x; i < n; i += kBlock) t += part[i];
  const double r = blur3_block_sum(t, s);
  if (threadIdx.x == 0) result[blockIdx.x] = r * scale;
}


// What closes a Lanczos half: the per-tile partials summed in a fixed order onto the
// scalar board, and the coefficients of the NEXT kernel formed from the board in double
// (IEEE division and square root: the values the host forms from the same sums).
//   which 0 (init):    coef[0..2] = rho_g / beta_0, rho_i / beta_0, 0
//   which 1 (after A): board[3 j + 1] = sum t^2, board[3 j + 2] = sum |grad y|^2;
//                      alpha = (tt + rho_g gg) / |y_j|^2 + rho_i;
//                      coef[4] = 1 / beta_j, coef[5] = -alpha / beta_j
//   which 2 (after B): board[3 j + 3] = sum y_new^2 = beta_{j+1}^2;
//                      coef[0..2] = rho_g / beta_{j+1}, rho_i / beta_{j+1}, -beta_{j+1} / beta_j
template <typename T>
__global__ __launch_bounds__(kBlock) void k_blur3_lanczos_final(
    const double *part, int n, int which, double *board, int j, double rho_g, double rho_i,
    T *coef) {
  __shared__ double s[2][kBlock / 64];
  const int nsum = which == 1 ? 2 : (which == 2 ? 1 : 0);
  // (the board value the coefficients need, requested ahead of the sums)
  const double nb2_j = (which != 0 && threadIdx.x == 0) ? board[3 * j] : 0.0;
  double r[2] = {0.0, 0.0};
  for (int a = 0; a < nsum; ++a) {
    double t = 0.0;
    for (int i = threadIdx.x; i < n; i += kBlock) t += part[(size_t)a * n + i];
    r[a] = blur3_block_sum(t, s[a]);
  }
  if (threadIdx.x != 0) return;
  if (which == 0) {
    const double b0 = sqrt(board[0]);
    coef[0] = (T)(rho_g / b0);
    coef[1] = (T)(rho_i / b0);
    coef[2] = T(0);
  } else if (which == 1) {
    board[3 * j + 1] = r[0];
    board[3 * j + 2] = r[1];
    const double nb2 = nb2_j;
    const double alpha = (r[0] + rho_g * r[1]) / nb2 + rho_i;
    const double beta = sqrt(nb2);
    coef[4] = (T)(1.0 / beta);
    coef[5] = (T)(-alpha / beta);
  } else {
    board[3 * j + 3] = r[0];
    const double bn = sqrt(r[0]), beta = sqrt(nb2_j);
    coef[0] = (T)(rho_g / bn);
    coef[1] = (T)(rho_i / bn);
    coef[2] = (T)(-bn / beta);
  }
}

// The first half taken by the EPI 2 kernel (its two sums already reduced): what
// k_blur3_lanczos_final does for which == 1, from sums2 = { sum t^2, sum |grad y|^2 }.
template <typename T>
__global__ void k_blur3_lanczos_from_sums(const double *sums2, double *board, int j,
                                          double rho_g, double rho_i, T *coef) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double r0 = sums2[0], r1 = sums2[1];
  board[3 * j + 1] = r0;
  board[3 * j + 2] = r1;
  const double nb2 = board[3 * j];
  const double alpha = (r0 + rho_g * r1) / nb2 + rho_i;
  const double beta = sqrt(nb2);
  coef[4] = (T)(1.0 / beta);
  coef[5] = (T)(-alpha / beta);
}

template <typename T>
struct LanczosArgs {
  const T *aux1, *aux2;
  T *aux_out, *coef;
  double *board;
  int step;
  double rho_g, rho_i;
};

// LDS-DMA staged kernel: tiles of kDmaLxb lanes per row whatever the row length;
// returns -2 when it does not apply.  EPI 1 (out = io, in place): io = ca * blur(x) +
// cb * io and *result = sum of squares of the new io (part: >= tiles doubles).  EPI 2:
// see the kernel (result[0 .. 1], part: >= 2 * tiles doubles).
template <typename T, int VEC, int NT, int NWD, int EPI = 0>
int launch_blur3_dma(const T *x, T *out, int64_t nz, int64_t ny, int64_t nx,
                     const Taps<T> &tz, const Taps<T> &ty, const Taps<T> &tx,
                     hipStream_t st, double ca = 1.0, double cb = 0.0, double cc = 0.0,
                     double *result = nullptr, double *part = nullptr,
                     int64_t part_doubles = 0, const LanczosArgs<T> *lz = nullptr) {
  constexpr int R = NT / 2;
  constexpr int NBH = (R + VEC - 1) / VEC;
  constexpr int dl = kDmaLxb;
  constexpr int dtyr = (NWD * 64) / dl;
  constexpr int frows = dtyr + 2 * R;
  constexpr int npieces = (frows * (dl + 2 * NBH) + 63) / 64;
  // own-position tiles staged beside the raw tiles: EPI 1 two (old io, double buffered),
  // EPI 3 two (y_prev, double buffered), EPI 4 two (q0 and y, one buffer each)
  constexpr int otiles = EPI == 1 ? 2 : (EPI == 3 ? 2 : (EPI == 4 ? 2 : (EPI == 5 ? 2 : 0)));
  // (EPI 6: one own-position tile and the halo'd tile of y, whole 1-KiB pieces)
  constexpr size_t ovecs =
      EPI == 6 ? (size_t)dtyr * dl + (blur3_epi6_mini<T, VEC, NT, NWD>()
                                          ? (size_t)NWD * 128
                                          : (((size_t)(dtyr + 2) * (dl + 2) + 63) / 64) * 64)
               : (size_t)otiles * dtyr * dl;
  constexpr size_t lds0 = (3 * (size_t)npieces * 64 + 2 * (size_t)frows * dl + ovecs) * 16;
  constexpr size_t lds_patch = 3 * (size_t)((frows * 4 + 63) / 64) * 16 * 16;   // (RAG)
  if constexpr (lds0 > 160 * 1024 || ((EPI == 2 || EPI == 3) && sizeof(T) == 8 && NT >= 15) ||
                (EPI >= 3 && (NT < 5 || NT >= 15 ||
                              (sizeof(T) == 8 && NT >= (EPI == 6 ? 13 : 11))))) {
    static_assert(EPI != 0, "LDS-DMA blur tile does not fit");
    // (no room for the io tiles; the difference sums of 15 / 17 taps in double -- and the
    // first Lanczos half at 15 / 17 taps, in double from 11 taps on (the lean second half:
    // from 13 on) -- would need more than the 128 registers of a 16-wave workgroup)
    return -2;
  } else {
  if (dtyr < 2 * R) return -2;
  // rows that are not whole vectors, or operands off the 16-byte grid: the RAG form
  // (it wants a row at least as long as a raw tile row: one row end per window)
  const bool rag = nx % VEC != 0 || ((reinterpret_cast<uintptr_t>(x) |
                                      reinterpret_cast<uintptr_t>(out)) & 15u);
  if (rag && (!nsol_blur3_dma_rag || nx < (int64_t)(dl + 2 * NBH) * VEC)) return -2;
  if constexpr (EPI >= 3) {
    if (rag || !lz) return -2;                     // (no ragged form of the Lanczos halves)
    if ((reinterpret_cast<uintptr_t>(lz->aux1) | reinterpret_cast<uintptr_t>(lz->aux2) |
         reinterpret_cast<uintptr_t>(lz->aux_out)) & 15u)
      return -2;
  }
  const size_t lds = lds0 + (rag ? lds_patch : 0);
  if (lds > 160 * 1024) return -2;
  const int64_t nxv = (nx + VEC - 1) / VEC;
  const int64_t dntx = (nxv + dl - 1) / dl, dnty = (ny + dtyr - 1) / dtyr;
  if (ny * nx >= ((int64_t)1 << 31)) return -2;          // 32-bit offsets in a plane
  if (nz >= ((int64_t)1 << 30)) return -2;               // 32-bit plane indices
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
  if (nsol_blur3_zchunk > 0) zchunk = nsol_blur3_zchunk < nz ? nsol_blur3_zchunk : nz;
  const int64_t nzc = (nz + zchunk - 1) / zchunk;
  const int64_t tiles = dntx * dnty * nzc;
  if (tiles >= ((int64_t)1 << 28)) return -2;
  if (EPI != 0 && tiles * ((EPI == 2 || EPI == 3) ? 2 : 1) > part_doubles) return -2;
  const int per_xcd = (int)((tiles + 7) / 8);
  bool iso = true;
  for (int t = 0; t < NT; ++t) iso = iso && tz.w[t] == tx.w[t] && ty.w[t] == tx.w[t];
  // (the difference sums on ragged rows at 17 taps would spill: the caller takes
  // nsol_tk1_grad_norm_* beside the epilogue form there)
  constexpr bool RG = !(EPI == 2 && NT >= 17) && EPI < 3;
  if (rag && !RG) return -2;
  auto kern = rag ? (iso ? k_blur3_dma<T, VEC, NT, NWD, true, EPI, RG>
                         : k_blur3_dma<T, VEC, NT, NWD, false, EPI, RG>)
                  : (iso ? k_blur3_dma<T, VEC, NT, NWD, true, EPI, false>
                         : k_blur3_dma<T, VEC, NT, NWD, false, EPI, false>);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) { (void)hipGetLastError(); return -2; }
  }
  if constexpr (EPI == 5) {
    hipLaunchKernelGGL(kern, dim3((unsigned)(per_xcd * 8)), dim3(NWD * 64), lds, st, x, out,
                       nz, ny, nx, tz, ty, tx, (int)dntx, (int)dnty, (int)nzc, (int)zchunk,
                       per_xcd, (T)ca, (T)cb, (T)cc, part, lz->aux1, (const T *)nullptr,
                       (T *)nullptr, (const T *)nullptr);
    hipLaunchKernelGGL(k_blur3_epi_final, dim3(1), dim3(kBlock), 0, st, part, (int)tiles,
                       result, 0.5);                     // (1/2 sum rho, as nsol_loss_*)
    return launch_status();
  }
  if constexpr (EPI >= 3) {
    hipLaunchKernelGGL(kern, dim3((unsigned)(per_xcd * 8)), dim3(NWD * 64), lds, st, x, out,
                       nz, ny, nx, tz, ty, tx, (int)dntx, (int)dnty, (int)nzc, (int)zchunk,
                       per_xcd, (T)ca, (T)cb, (T)cc, part, lz->aux1, lz->aux2, lz->aux_out,
                       (const T *)lz->coef);
    hipLaunchKernelGGL(k_blur3_lanczos_final<T>, dim3(1), dim3(kBlock), 0, st, part,
                       (int)tiles, EPI == 3 ? 1 : 2, lz->board, lz->step, lz->rho_g,
                       lz->rho_i, lz->coef);
    return launch_status();
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(per_xcd * 8)), dim3(NWD * 64), lds, st, x, out,
                     nz, ny, nx, tz, ty, tx, (int)dntx, (int)dnty, (int)nzc, (int)zchunk,
                     per_xcd, (T)ca, (T)cb, (T)cc, part, (const T *)nullptr,
                     (const T *)nullptr, (T *)nullptr, (const T *)nullptr);
  if constexpr (EPI == 2) {
    if (lz) {                    // the first Lanczos half: sums -> board, coefficients
      hipLaunchKernelGGL(k_blur3_lanczos_final<T>, dim3(1), dim3(kBlock), 0, st, part,
                         (int)tiles, 1, lz->board, lz->step, lz->rho_g, lz->rho_i, lz->coef);
      return launch_status();
    }
  }
  if (EPI != 0)
    hipLaunchKernelGGL(k_blur3_epi_final, dim3(EPI == 2 ? 2 : 1), dim3(kBlock), 0, st, part,
                       (int)tiles, result, 1.0);
  return launch_status();
  }
}

template <typename T>
int blur3_dma_dispatch(const T *x, T *out, int64_t nz, int64_t ny, int64_t nx,
                       const Taps<T> &tz, const Taps<T> &ty, const Taps<T> &tx, int ntaps,
                       int epi, double ca, double cb, double cc, double *result,
                       double *part, int64_t part_doubles, hipStream_t st,
                       const LanczosArgs<T> *lz = nullptr) {
  constexpr int VEC = 16 / sizeof(T);
#define NSOL_B3D_CASE(N)                                                                  \
  case N:                                                                                 \
    return epi == 2 ? launch_blur3_dma<T, VEC, N, 16, 2>(x, out, nz, ny, nx, tz, ty, tx, st, \
                                                         ca, cb, cc, result, part,         \
                                                         part_doubles, lz)                 \
           : epi    ? launch_blur3_dma<T, VEC, N, 16, 1>(x, out, nz, ny, nx, tz, ty, tx, st, \
                                                         ca, cb, 0.0, result, part,        \
                                                         part_doubles)                     \
                    : launch_blur3_dma<T, VEC, N, 16, 0>(x, out, nz, ny, nx, tz, ty, tx, st);
  switch (ntaps) {
    NSOL_B3D_CASE(3) NSOL_B3D_CASE(5) NSOL_B3D_CASE(7) NSOL_B3D_CASE(9)
    NSOL_B3D_CASE(11) NSOL_B3D_CASE(13) NSOL_B3D_CASE(15) NSOL_B3D_CASE(17)
    default: return -2;
  }
#undef NSOL_B3D_CASE
}

template <typename T, int EPI>
int blur3_lanczos_dispatch(const T *x, T *out, int64_t nz, int64_t ny, int64_t nx,
                           const Taps<T> &tz, const Taps<T> &ty, const Taps<T> &tx, int ntaps,
                           const LanczosArgs<T> &lz, double *part, int64_t part_doubles,
                           hipStream_t st) {
  constexpr int VEC = 16 / sizeof(T);
  // (the squared weights of the difference sums: unit spacing)
#define NSOL_B3L_CASE(N)                                                                    \
  case N:                                                                                   \
    return launch_blur3_dma<T, VEC, N, 16, EPI>(x, out, nz, ny, nx, tz, ty, tx, st, 1.0, 1.0, \
                                                1.0, nullptr, part, part_doubles, &lz);
  switch (ntaps) {
    NSOL_B3L_CASE(5) NSOL_B3L_CASE(7) NSOL_B3L_CASE(9) NSOL_B3L_CASE(11) NSOL_B3L_CASE(13)
    NSOL_B3L_CASE(15) NSOL_B3L_CASE(17)
    default: return -2;
  }
#undef NSOL_B3L_CASE
}

template <typename T>
int blur3_loss_dispatch(const T *x, T *out, int64_t nz, int64_t ny, int64_t nx,
                        const Taps<T> &tz, const Taps<T> &ty, const Taps<T> &tx, int ntaps,
                        const LanczosArgs<T> &lz, int loss, double s2, double gm, double *result,
                        double *part, int64_t part_doubles, hipStream_t st) {
  constexpr int VEC = 16 / sizeof(T);
#define NSOL_B3S_CASE(N)                                                                    \
  case N:                                                                                   \
    return launch_blur3_dma<T, VEC, N, 16, 5>(x, out, nz, ny, nx, tz, ty, tx, st, s2, gm,   \
                                              (double)loss, result, part, part_doubles, &lz);
  switch (ntaps) {
    NSOL_B3S_CASE(5) NSOL_B3S_CASE(7) NSOL_B3S_CASE(9) NSOL_B3S_CASE(11) NSOL_B3S_CASE(13)
    NSOL_B3S_CASE(15) NSOL_B3S_CASE(17)
    default: return -2;
  }
#undef NSOL_B3S_CASE
}

}  // namespace

#define NSOL_B3L_DEF(T)                                                                      \
  int blur3_lanczos_a(const T *y, const T *y_prev, T *t, T *q0, int64_t nz, int64_t ny,      \
                      int64_t nx, const Taps<T> &tz, const Taps<T> &ty, const Taps<T> &tx,   \
                      int ntaps, double rho_g, double rho_i, double *board, int step, T *coef, \
                      double *part, int64_t part_doubles, hipStream_t st) {                  \
    const LanczosArgs<T> lz{y_prev, nullptr, q0, coef, board, step, rho_g, rho_i};           \
    return blur3_lanczos_dispatch<T, 3>(y, t, nz, ny, nx, tz, ty, tx, ntaps, lz, part,       \
                                        part_doubles, st);                                   \
  }                                                                                          \
  int blur3_lanczos_b(const T *t, const T *q0, const T *y, T *y_new, int64_t nz, int64_t ny, \
                      int64_t nx, const Taps<T> &tz, const Taps<T> &ty, const Taps<T> &tx,   \
                      int ntaps, double rho_g, double rho_i, double *board, int step, T *coef, \
                      double *part, int64_t part_doubles, hipStream_t st) {                  \
    const LanczosArgs<T> lz{q0, y, nullptr, coef, board, step, rho_g, rho_i};                \
    return blur3_lanczos_dispatch<T, 4>(t, y_new, nz, ny, nx, tz, ty, tx, ntaps, lz, part,   \
                                        part_doubles, st);                                   \
  }                                                                                          \
  int blur3_loss_epilogue(const T *x, const T *b, T *g, int64_t nz, int64_t ny, int64_t nx,  \
                          const Taps<T> &tz, const Taps<T> &ty, const Taps<T> &tx, int ntaps, \
                          int loss, double s2, double gm, double *result, double *part,      \
                          int64_t part_doubles, hipStream_t st) {                            \
    const LanczosArgs<T> lz{b, nullptr, nullptr, nullptr, nullptr, 0, 0.0, 0.0};             \
    return blur3_loss_dispatch<T>(x, g, nz, ny, nx, tz, ty, tx, ntaps, lz, loss, s2, gm,     \
                                  result, part, part_doubles, st);                           \
  }                                                                                          \
  int blur3_lanczos_b2(const T *t, const T *y, const T *y_prev, T *y_new, int64_t nz,        \
                       int64_t ny, int64_t nx, const Taps<T> &tz, const Taps<T> &ty,         \
                       const Taps<T> &tx, int ntaps, double rho_g, double rho_i,             \
                       double *board, int step, T *coef, double *part, int64_t part_doubles, \
                       hipStream_t st) {                                                     \
    /* y_new == NULL: the sum of squares alone (aux_out doubles as that flag; the kernel's  \
       stores then go nowhere and `out` only has to be a valid, aligned address) */         \
    const LanczosArgs<T> lz{y, y_prev, y_new ? nullptr : const_cast<T *>(t), coef, board,    \
                            step, rho_g, rho_i};                                             \
    return blur3_lanczos_dispatch<T, 6>(t, y_new ? y_new : const_cast<T *>(t), nz, ny, nx,   \
                                        tz, ty, tx, ntaps, lz, part, part_doubles, st);      \
  }                                                                                          \
  int blur3_lanczos_a2_close(const double *sums2, double *board, int step, double rho_g,     \
                             double rho_i, T *coef, hipStream_t st) {                        \
    hipLaunchKernelGGL(k_blur3_lanczos_from_sums<T>, dim3(1), dim3(64), 0, st, sums2, board, \
                       step, rho_g, rho_i, coef);                                            \
    return launch_status();                                                                  \
  }                                                                                          \
  int blur3_lanczos_init(double *board, T *coef, double rho_g, double rho_i,                 \
                         hipStream_t st) {                                                   \
    hipLaunchKernelGGL(k_blur3_lanczos_final<T>, dim3(1), dim3(kBlock), 0, st, nullptr, 0, 0, \
                       board, 0, rho_g, rho_i, coef);                                        \
    return launch_status();                                                                  \
  }

}  // namespace nsol_blur3
#endif  // NSOL_BLUR3_DMA_IMPL
