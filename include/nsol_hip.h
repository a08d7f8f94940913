/*
 * nsol_hip.h -- C ABI of libnsol_hip.so: MI355X (gfx950) kernels for the
 * per-iteration hot path of NSoL's PrimalDualSolver.run() and
 * ADMMLinearSolver.run().
 *
 * The reference (gift-surg/NSoL v0.1.14) is pure Python and has no FFI; its
 * operator interface is a set of Python callables on flat arrays.  Each entry
 * point below replaces the NumPy/SciPy call sites named in its comment
 * (paths relative to the reference root).  The Python host package
 * `nsol_amd` binds these symbols with ctypes.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller, except where the
 *    name ends in `_host`;
 *  - volumes are C-contiguous [nz][ny][nx] (x fastest); 2-D: nz = 1;
 *    1-D: nz = ny = 1.  `ndim` (1..3) says how many gradient components exist;
 *  - a gradient field is `ndim` stacked volumes [x-comp; y-comp; z-comp]
 *    (reference linear_operators.py:140 np.concatenate on axis 0);
 *  - wx, wy, wz are the INVERSE spacings 1/h of the x, y, z axes
 *    (reference kernels.py:102-112,160-190,240-286 scale taps by 1/spacing);
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *  - functions never allocate device memory, never synchronise, never throw
 *    (nsol_pd_persist_run_* works in a caller-owned scratch buffer);
 *    they return 0 on success, a positive hipError_t on a HIP failure and
 *    NSOL_EINVAL (-1) on bad arguments.  They keep no state between calls with
 *    two exceptions, neither of which changes results: (1) the experiment
 *    knobs of nsol_hip_set_param / _pd2 / _pdk / _conv / _lb (name, value) -- tile
 *    shapes, z-chunk lengths; (2) nsol_pd_fusedk_iter_* / nsol_pd_run_* keep a
 *    process-wide, mutex-protected table of footprint plans keyed by (current
 *    device ordinal, element size, depth K, nz, ny, nx), each holding a few
 *    hipEvents of launches still in flight (created on that device, read back
 *    with hipEventQuery, destroyed when read or by the knob "pdk_forget").
 *    The kernel variants of one shape (TV / Huber, l1 / l2, unit / non-unit
 *    spacing) share a plan: it fixes the footprint geometry only.  The
 *    package is written for one process per GPU; a process that drives several
 *    devices gets one plan per device;
 *  - suffix _f32 / _f64 = element type of all volume arguments.  Scalars are
 *    always passed as double and rounded to the element type inside.
 */
#ifndef NSOL_HIP_H
#define NSOL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NSOL_HIP_ABI_VERSION 1
#define NSOL_EINVAL (-1)

/* boundary modes of scipy.ndimage.convolve as used by
 * linear_operators.py:60-68 (mode="wrap") and :98-106 (mode="constant") */
#define NSOL_MODE_CONSTANT 0
#define NSOL_MODE_WRAP 1
#define NSOL_MODE_NEAREST 2
#define NSOL_MODE_REFLECT 3 /* d c b a | a b c d | d c b a */
#define NSOL_MODE_MIRROR 4  /* d c b | a b c d | c b a   */

/* flags of the primal-dual kernels */
#define NSOL_PD_REG_TV 0    /* proximal_operators.py:138-140 prox_tv_conj   */
#define NSOL_PD_REG_HUBER 1 /* proximal_operators.py:156-159 prox_huber_conj */
#define NSOL_PD_DATA_L2 0   /* proximal_operators.py:117-120 prox_ell2_denoising */
#define NSOL_PD_DATA_L1 2   /* proximal_operators.py:95-98  prox_ell1_denoising */
/* nsol_pd_run_* only: the caller can take the final primal iterate from x_alt (see
 * there) -- no copy back into x after an odd number of multi-iteration launches */
#define NSOL_PD_RUN_X_MAY_SWAP 0x100

/* loss ids, loss_functions.py:251-266 */
#define NSOL_LOSS_LINEAR 0
#define NSOL_LOSS_SOFT_L1 1
#define NSOL_LOSS_HUBER 2
#define NSOL_LOSS_CAUCHY 3
#define NSOL_LOSS_ARCTAN 4

int nsol_hip_abi_version(void);
/* number of doubles a reduction workspace must hold (see *_ws arguments) */
int nsol_hip_reduce_ws_doubles(void);

/* ---------------------------------------------------------------------- *
 * Finite differences
 * ---------------------------------------------------------------------- */
/* K = grad: g[a] = D_a x, (D_a x)[i] = x[i+e_a]*w_a + x[i]*(-w_a), x := 0 past
 * the last index.  Replaces linear_operators.py:121-144 (`grad`, three
 * ndimage.convolve calls + np.concatenate). */
int nsol_grad_f32(const float *x, float *g, int ndim, int64_t nz, int64_t ny,
                  int64_t nx, double wx, double wy, double wz, void *stream);
int nsol_grad_f64(const double *x, double *g, int ndim, int64_t nz, int64_t ny,
                  int64_t nx, double wx, double wy, double wz, void *stream);
/* K^T: out[i] = sum_a p_a[i]*(-w_a) + p_a[i-e_a]*w_a, p := 0 before the first
 * index.  Replaces linear_operators.py:158-169 (_get_adjoint_gradient_operator). */
int nsol_grad_adj_f32(const float *p, float *out, int ndim, int64_t nz,
                      int64_t ny, int64_t nx, double wx, double wy, double wz,
                      void *stream);
int nsol_grad_adj_f64(const double *p, double *out, int ndim, int64_t nz,
                      int64_t ny, int64_t nx, double wx, double wy, double wz,
                      void *stream);
/* out = x - tau * grad_adj(p): the argument of prox_f (primal_dual_solver.py:246-248)
 * in one pass, for a prox_f that is not one of the fused ones (e.g.
 * prox_linear_least_squares, proximal_operators.py:43-78).  out may alias x. */
int nsol_grad_adj_axpy_f32(const float *p, const float *x, float *out, int ndim,
                           int64_t nz, int64_t ny, int64_t nx, double wx, double wy,
                           double wz, double tau, void *stream);
int nsol_grad_adj_axpy_f64(const double *p, const double *x, double *out, int ndim,
                           int64_t nz, int64_t ny, int64_t nx, double wx, double wy,
                           double wz, double tau, void *stream);
/* out = a + theta * (a - b): the over-relaxation xbar = x_new + theta (x_new - x)
 * (primal_dual_solver.py:252-253) with the reference's rounding.  out may alias b. */
int nsol_extrapolate_f32(float *out, const float *a, const float *b, double theta,
                         int64_t n, void *stream);
int nsol_extrapolate_f64(double *out, const double *a, const double *b, double theta,
                         int64_t n, void *stream);
/* single-axis D_a (adjoint = 0) or D_a^T (adjoint = 1); dir 0 = x, 1 = y, 2 = z.
 * Replaces linear_operators.py:98-106, 193-247 (get_d{x,y,z}_operators). */
int nsol_diff_axis_f32(const float *x, float *out, int dir, int adjoint,
                       int64_t nz, int64_t ny, int64_t nx, double w,
                       void *stream);
int nsol_diff_axis_f64(const double *x, double *out, int dir, int adjoint,
                       int64_t nz, int64_t ny, int64_t nx, double w,
                       void *stream);

/* ---------------------------------------------------------------------- *
 * Convolution (Gaussian blur A = A^T and user kernels)
 * ---------------------------------------------------------------------- */
/* 1-D correlation along one ARRAY axis (0 = z, 1 = y, 2 = x):
 *   out[i] = sum_{t<ntaps} taps_host[t] * x[i + t - centre]   (index mapped by `mode`).
 * Three calls implement the separable Gaussian of linear_operators.py:82-86 /
 * kernels.py:198-238 (diagonal covariance).  taps_host: HOST pointer, ntaps <= 129;
 * the taps are copied into kernel arguments.  x and out must not alias. */
int nsol_corr_axis_f32(const float *x, float *out, int axis, int64_t nz,
                       int64_t ny, int64_t nx, const double *taps_host,
                       int ntaps, int centre, int mode, void *stream);
int nsol_corr_axis_f64(const double *x, double *out, int axis, int64_t nz,
                       int64_t ny, int64_t nx, const double *taps_host,
                       int ntaps, int centre, int mode, void *stream);
/* Separable periodic ("wrap") 3-D correlation in one pass over memory: the
 * Gaussian blur A = A^T of linear_operators.py:82-86 on a volume, 8 bytes per
 * voxel instead of 24 for three nsol_corr_axis_* passes.  taps_* are HOST arrays
 * of `ntaps` doubles (odd, centre in the middle, the same count on every axis).
 * Passes run x, y, z.  Rows need not be whole 16-byte vectors and x / out need
 * only be element-aligned (symmetric taps and rows of at least 16 + 2 * ceil(
 * (ntaps / 2) / VEC) vectors of VEC = 16 / sizeof(T) elements; other ragged or
 * off-grid cases take a slower kernel or -2).  Returns -2 (nothing launched)
 * when no one-pass kernel applies (even tap counts or more than 17, short ragged
 * rows, planes of more than 3 GiB): use the per-axis passes then. */
int nsol_corr3_wrap_f32(const float *x, float *out, int64_t nz, int64_t ny,
                        int64_t nx, const double *taps_z, const double *taps_y,
                        const double *taps_x, int ntaps, void *stream);
int nsol_corr3_wrap_f64(const double *x, double *out, int64_t nz, int64_t ny,
                        int64_t nx, const double *taps_z, const double *taps_y,
                        const double *taps_x, int ntaps, void *stream);
/* The same blur with LSMR's top-block update as its epilogue
 * (tikhonov_linear_solver.py:226-274 driving SciPy's lsmr.py:320-336,
 * `u = A v - alpha u`): io = ca * (A x) + cb * io in place and *result = the sum
 * of squares of the new io (double), without A x going to memory.  ws: device
 * scratch of ws_doubles doubles (one per tile; nsol_hip_reduce_ws_doubles()
 * suffices up to 2048^3).  Returns -2 (nothing launched) where the LDS-DMA staged
 * kernel does not apply: asymmetric taps, even / > 17 taps, ragged rows shorter
 * than the bound above, 17 taps in double (no LDS left for the io tiles). */
int nsol_corr3_wrap_axpby_f32(const float *x, float *io, int64_t nz, int64_t ny,
                              int64_t nx, const double *taps_z, const double *taps_y,
                              const double *taps_x, int ntaps, double ca, double cb,
                              double *result, double *ws, int64_t ws_doubles,
                              void *stream);
int nsol_corr3_wrap_axpby_f64(const double *x, double *io, int64_t nz, int64_t ny,
                              int64_t nx, const double *taps_z, const double *taps_y,
                              const double *taps_x, int ntaps, double ca, double cb,
                              double *result, double *ws, int64_t ws_doubles,
                              void *stream);
/* The same blur with the two sums of a Lanczos step on the normal equations
 * A^T A + rho grad^T grad (how nsol_amd runs the LSMR solve of
 * tikhonov_linear_solver.py:146-158 with a first-order Tikhonov regulariser,
 * nsol_amd/lsmr.py): out = A x, result[0] = sum (A x)^2 and result[1] =
 * sum |grad x|^2 of the INPUT -- forward differences with weights wx, wy, wz (the
 * inverse spacings) and zero behind the last voxel of an axis, exactly
 * nsol_tk1_grad_norm_*'s sum, taken from the tiles of x the blur stages anyway
 * instead of a second pass over x.  ws: >= 2 doubles per tile.  Returns -2 (nothing
 * launched) where nsol_corr3_wrap_axpby_* does, for 15 / 17 taps in double and
 * for ragged rows or off-grid operands at 17 taps. */
int nsol_corr3_wrap_norms_f32(const float *x, float *out, int64_t nz, int64_t ny,
                              int64_t nx, const double *taps_z, const double *taps_y,
                              const double *taps_x, int ntaps, double wx, double wy,
                              double wz, double *result, double *ws, int64_t ws_doubles,
                              void *stream);
int nsol_corr3_wrap_norms_f64(const double *x, double *out, int64_t nz, int64_t ny,
                              int64_t nx, const double *taps_z, const double *taps_y,
                              const double *taps_x, int ntaps, double wx, double wy,
                              double wz, double *result, double *ws, int64_t ws_doubles,
                              void *stream);
/* LSMR as Lanczos on the normal equations A'A + rho B'B (replaces SciPy's lsmr.py:320-413
 * behind tikhonov_linear_solver.py:146-158) with BOTH halves of a step taken by the
 * one-pass blur A = A' (linear_operators.py:60-86), unit spacing, B = gradient (rho_grad)
 * or identity (rho_ident):
 *   _a:  t = A y with sum t^2 and sum |grad y|^2, and
 *        q0 = c1 K'K y + c0 y + c2 y_prev           (y_prev NULL in the first step)
 *   _b:  y_new = ca A t + q0 + cy y with sum y_new^2
 * board (device, 3 * (steps + 1) doubles; the caller stores |y_0|^2 in board[0]):
 * board[3 j] = |y_j|^2, [3 j + 1] = |A y_j|^2, [3 j + 2] = |grad y_j|^2.  coef (device,
 * 8 elements of the array type): c1, c0, c2 at [0..2], ca, cy at [4..5]; _init writes
 * [0..2] for step 0 from board[0], every _a writes [4..5], every _b [0..2] for the next
 * step -- steps follow each other on the stream without the host reading a scalar.
 * q0 equals nsol_tk1_lanczos_* (c_g = 0) and y_new nsol_lincomb3_* of the plain blur bit
 * for bit.  ws: >= 2 doubles per tile of scratch (nsol_hip_reduce_ws_doubles()).
 * Return -2 (nothing launched) where the form does not apply: rows not a multiple of 16
 * bytes, unaligned arrays, fewer than 5 or more than 13 (float64: 9) taps, taps not
 * symmetric. */
int nsol_corr3_wrap_lanczos_init_f32(double *board, float *coef, double rho_grad,
                                     double rho_ident, void *stream);
int nsol_corr3_wrap_lanczos_init_f64(double *board, double *coef, double rho_grad,
                                     double rho_ident, void *stream);
int nsol_corr3_wrap_lanczos_a_f32(const float *y, const float *y_prev, float *t, float *q0,
                                  int64_t nz, int64_t ny, int64_t nx, const double *taps_z,
                                  const double *taps_y, const double *taps_x, int ntaps,
                                  double rho_grad, double rho_ident, double *board, int step,
                                  float *coef, double *ws, int64_t ws_doubles, void *stream);
int nsol_corr3_wrap_lanczos_a_f64(const double *y, const double *y_prev, double *t,
                                  double *q0, int64_t nz, int64_t ny, int64_t nx,
                                  const double *taps_z, const double *taps_y,
                                  const double *taps_x, int ntaps, double rho_grad,
                                  double rho_ident, double *board, int step, double *coef,
                                  double *ws, int64_t ws_doubles, void *stream);
int nsol_corr3_wrap_lanczos_b_f32(const float *t, const float *q0, const float *y,
                                  float *y_new, int64_t nz, int64_t ny, int64_t nx,
                                  const double *taps_z, const double *taps_y,
                                  const double *taps_x, int ntaps, double rho_grad,
                                  double rho_ident, double *board, int step, float *coef,
                                  double *ws, int64_t ws_doubles, void *stream);
int nsol_corr3_wrap_lanczos_b_f64(const double *t, const double *q0, const double *y,
                                  double *y_new, int64_t nz, int64_t ny, int64_t nx,
                                  const double *taps_z, const double *taps_y,
                                  const double *taps_x, int ntaps, double rho_grad,
                                  double rho_ident, double *board, int step, double *coef,
                                  double *ws, int64_t ws_doubles, void *stream);

/* The leaner pair of halves (same board, same coefficients, same results bit for bit):
 * _a2: t = A y with sum t^2 and sum |grad y|^2 onto the board (the blur of
 * nsol_corr3_wrap_norms_*) and coef[4..5] for the second half -- no q0;  _b2:
 * y_new = ca A t + (c1 K'K y + c0 y + c2 y_prev) + cy y with sum y_new^2, the step's
 * K'K y formed from a halo'd tile of y inside the kernel (y_prev may be NULL).  25 B per
 * voxel and step instead of 33.  y_new may be NULL: only its sum of squares (the next
 * beta) is wanted -- the last step of a solve, whose vector nobody reads.  Return -2 as
 * _a / _b do. */
int nsol_corr3_wrap_lanczos_a2_f32(const float *y, float *t, int64_t nz, int64_t ny,
                                   int64_t nx, const double *taps_z, const double *taps_y,
                                   const double *taps_x, int ntaps, double rho_grad,
                                   double rho_ident, double *board, int step, float *coef,
                                   double *ws, int64_t ws_doubles, void *stream);
int nsol_corr3_wrap_lanczos_a2_f64(const double *y, double *t, int64_t nz, int64_t ny,
                                   int64_t nx, const double *taps_z, const double *taps_y,
                                   const double *taps_x, int ntaps, double rho_grad,
                                   double rho_ident, double *board, int step, double *coef,
                                   double *ws, int64_t ws_doubles, void *stream);
int nsol_corr3_wrap_lanczos_b2_f32(const float *t, const float *y, const float *y_prev,
                                   float *y_new, int64_t nz, int64_t ny, int64_t nx,
                                   const double *taps_z, const double *taps_y,
                                   const double *taps_x, int ntaps, double rho_grad,
                                   double rho_ident, double *board, int step, float *coef,
                                   double *ws, int64_t ws_doubles, void *stream);
int nsol_corr3_wrap_lanczos_b2_f64(const double *t, const double *y, const double *y_prev,
                                   double *y_new, int64_t nz, int64_t ny, int64_t nx,
                                   const double *taps_z, const double *taps_y,
                                   const double *taps_x, int ntaps, double rho_grad,
                                   double rho_ident, double *board, int step, double *coef,
                                   double *ws, int64_t ws_doubles, void *stream);

/* The blur with the data term of the robust-loss objective as its epilogue
 * (tikhonov_linear_solver.py:201-208: `residual = A(x) - b`, `loss(residual^2)`,
 * `A_adj(gradient_loss * residual)`): g = rho'(r^2) r for r = A x - b and
 * *result = 1/2 sum rho(r^2), what nsol_corr3_wrap_* followed by
 * nsol_loss_residual_cost_grad_* produce (g bit for bit, the sum in another order)
 * without A x going to memory.  loss: NSOL_LOSS_LINEAR / _SOFT_L1 / _HUBER.  ws: one
 * double per tile of scratch (nsol_hip_reduce_ws_doubles() suffices).  Returns -2
 * (nothing launched) for the other losses and where nsol_corr3_wrap_lanczos_a_* does. */
int nsol_corr3_wrap_loss_f32(const float *x, const float *b, float *g, int64_t nz,
                             int64_t ny, int64_t nx, const double *taps_z,
                             const double *taps_y, const double *taps_x, int ntaps,
                             int loss, double f_scale, double *result, double *ws,
                             int64_t ws_doubles, void *stream);
int nsol_corr3_wrap_loss_f64(const double *x, const double *b, double *g, int64_t nz,
                             int64_t ny, int64_t nx, const double *taps_z,
                             const double *taps_y, const double *taps_x, int ntaps,
                             int loss, double f_scale, double *result, double *ws,
                             int64_t ws_doubles, void *stream);
/* dense N-D correlation with DEVICE taps [kz][ky][kx] and centre (cz,cy,cx):
 *   out[i] = sum_t taps[t] * x[i + t - c].  Replaces linear_operators.py:60-68
 * (scipy.ndimage.convolve with an arbitrary kernel; the host flips the kernel
 * and derives the centre as ndimage does). */
int nsol_corr_dense_f32(const float *x, float *out, int64_t nz, int64_t ny,
                        int64_t nx, const float *taps, int kz, int ky, int kx,
                        int cz, int cy, int cx, int mode, void *stream);
int nsol_corr_dense_f64(const double *x, double *out, int64_t nz, int64_t ny,
                        int64_t nx, const double *taps, int kz, int ky, int kx,
                        int cz, int cy, int cx, int mode, void *stream);

/* ---------------------------------------------------------------------- *
 * Element-wise building blocks (n = number of elements; out may alias inputs)
 * ---------------------------------------------------------------------- */
/* out = a*x + b*y            (primal_dual_solver.py:243,246,253 axpys) */
int nsol_lincomb2_f32(float *out, double a, const float *x, double b,
                      const float *y, int64_t n, void *stream);
int nsol_lincomb2_f64(double *out, double a, const double *x, double b,
                      const double *y, int64_t n, void *stream);
/* out = a*x + b*y + c*z      (admm_linear_solver.py:208,222) */
int nsol_lincomb3_f32(float *out, double a, const float *x, double b,
                      const float *y, double c, const float *z, int64_t n,
                      void *stream);
int nsol_lincomb3_f64(double *out, double a, const double *x, double b,
                      const double *y, double c, const double *z, int64_t n,
                      void *stream);
/* out = x * a (divide = 0) or x / a (divide = 1)   (solver.py:37,117-118) */
int nsol_scale_f32(float *out, const float *x, double a, int divide, int64_t n,
                   void *stream);
int nsol_scale_f64(double *out, const double *x, double a, int divide,
                   int64_t n, void *stream);
/* out = min(max(x, lo), hi)  (tikhonov_linear_solver.py:142-143,156-158) */
int nsol_clip_f32(float *out, const float *x, double lo, double hi, int64_t n,
                  void *stream);
int nsol_clip_f64(double *out, const double *x, double lo, double hi,
                  int64_t n, void *stream);
/* out = (x / den) / max(1, |x / den|): prox_tv_conj (den = 1) and
 * prox_huber_conj (den = 1 + sigma*gamma), proximal_operators.py:138-159 */
int nsol_prox_dual_clamp_f32(float *out, const float *x, double den, int64_t n,
                             void *stream);
int nsol_prox_dual_clamp_f64(double *out, const double *x, double den,
                             int64_t n, void *stream);
/* out = (x + tau*bt) / (1 + tau), bt = b / x_scale precomputed;
 * proximal_operators.py:117-120 */
int nsol_prox_ell2_f32(float *out, const float *x, const float *bt, double tau,
                       int64_t n, void *stream);
int nsol_prox_ell2_f64(double *out, const double *x, const double *bt,
                       double tau, int64_t n, void *stream);
/* out = bt + max(|x - bt| - tau, 0) * sign(x - bt); proximal_operators.py:95-98 */
int nsol_prox_ell1_f32(float *out, const float *x, const float *bt, double tau,
                       int64_t n, void *stream);
int nsol_prox_ell1_f64(double *out, const double *x, const double *bt,
                       double tau, int64_t n, void *stream);

/* ---------------------------------------------------------------------- *
 * Reductions (deterministic two-stage; accumulate in double)
 * ws: device workspace of nsol_hip_reduce_ws_doubles() doubles;
 * result: device double[1] (dot, sumsq) -- read it back after the stream syncs.
 * ---------------------------------------------------------------------- */
/* result[0] = sum x[i]*y[i]   (np.linalg.norm in scipy lsmr.py:320-413) */
int nsol_dot_f32(const float *x, const float *y, int64_t n, double *result,
                 double *ws, void *stream);
int nsol_dot_f64(const double *x, const double *y, int64_t n, double *result,
                 double *ws, void *stream);

/* ---------------------------------------------------------------------- *
 * Primal-dual (Chambolle-Pock) iteration, primal_dual_solver.py:232-261
 * ---------------------------------------------------------------------- */
/* dual step: p_out = clamp((p_in + sigma * grad(xbar)) / hden)
 * (primal_dual_solver.py:242-243 with prox_tv_conj / prox_huber_conj);
 * p_in may be NULL (= 0, first iteration); p_out may alias p_in. */
int nsol_pd_dual_step_f32(const float *xbar, const float *p_in, float *p_out,
                          int ndim, int64_t nz, int64_t ny, int64_t nx,
                          double wx, double wy, double wz, double sigma,
                          double hden, void *stream);
int nsol_pd_dual_step_f64(const double *xbar, const double *p_in,
                          double *p_out, int ndim, int64_t nz, int64_t ny,
                          int64_t nx, double wx, double wy, double wz,
                          double sigma, double hden, void *stream);
/* primal step: u = x - tau*grad_adj(p); x_new = prox_f(u, tl); xbar = x_new +
 * theta*(x_new - x); x is updated in place (primal_dual_solver.py:246-256).
 * flags: NSOL_PD_DATA_L2 | NSOL_PD_DATA_L1; tl = tau*lambda. */
int nsol_pd_primal_step_f32(const float *p, float *x, float *xbar,
                            const float *bt, int ndim, int64_t nz, int64_t ny,
                            int64_t nx, double wx, double wy, double wz,
                            double tau, double tl, double theta, int flags,
                            void *stream);
int nsol_pd_primal_step_f64(const double *p, double *x, double *xbar,
                            const double *bt, int ndim, int64_t nz, int64_t ny,
                            int64_t nx, double wx, double wy, double wz,
                            double tau, double tl, double theta, int flags,
                            void *stream);
/* one whole iteration in a single pass over memory (11 words per voxel in 3-D):
 * reads xbar_in, x, bt, p_in; writes p_out, x (in place), xbar_out.
 * xbar_out must not alias xbar_in and p_out must not alias p_in (neighbouring
 * workgroups read the old values).  p_in may be NULL on the first iteration.
 * flags: NSOL_PD_REG_* | NSOL_PD_DATA_*. */
int nsol_pd_fused_iter_f32(const float *xbar_in, float *xbar_out, float *x,
                           const float *bt, const float *p_in, float *p_out,
                           int ndim, int64_t nz, int64_t ny, int64_t nx,
                           double wx, double wy, double wz, double sigma,
                           double hden, double tau, double tl, double theta,
                           int flags, void *stream);
int nsol_pd_fused_iter_f64(const double *xbar_in, double *xbar_out, double *x,
                           const double *bt, const double *p_in, double *p_out,
                           int ndim, int64_t nz, int64_t ny, int64_t nx,
                           double wx, double wy, double wz, double sigma,
                           double hden, double tau, double tl, double theta,
                           int flags, void *stream);
/* TWO iterations in a single pass over memory (temporal blocking; 11 words per
 * voxel for both).  *_2 arguments are HOST arrays of two doubles (iteration n,
 * n+1).  x is ping-pong as well (x_in != x_out): overlapping workgroup
 * footprints re-read neighbours' old x.  Returns -2 (and launches nothing) when
 * the kernel does not apply (ndim != 3, nx not a multiple of 16 bytes or
 * narrower than 64 vectors, unaligned pointers): call the one-iteration form
 * twice then.  Results are bit-identical to two nsol_pd_fused_iter_* calls. */
int nsol_pd_fused2_iter_f32(const float *xbar_in, float *xbar_out,
                            const float *x_in, float *x_out, const float *bt,
                            const float *p_in, float *p_out, int ndim,
                            int64_t nz, int64_t ny, int64_t nx, double wx,
                            double wy, double wz, const double *sigma_2,
                            const double *hden_2, const double *tau_2,
                            const double *tl_2, const double *theta_2,
                            int flags, void *stream);
int nsol_pd_fused2_iter_f64(const double *xbar_in, double *xbar_out,
                            const double *x_in, double *x_out,
                            const double *bt, const double *p_in,
                            double *p_out, int ndim, int64_t nz, int64_t ny,
                            int64_t nx, double wx, double wy, double wz,
                            const double *sigma_2, const double *hden_2,
                            const double *tau_2, const double *tl_2,
                            const double *theta_2, int flags, void *stream);
/* K = 2 or 3 iterations in a single pass over memory on 2-D tiled footprints
 * (temporal blocking of depth K; 11 words per voxel for all K).  sigma_k ...
 * theta_k are HOST arrays of k doubles (iterations n .. n+k-1).  Same aliasing
 * rules and -2 convention as nsol_pd_fused2_iter_*; results are bit-identical
 * to k calls of nsol_pd_fused_iter_*. */
int nsol_pd_fusedk_iter_f32(const float *xbar_in, float *xbar_out,
                            const float *x_in, float *x_out, const float *bt,
                            const float *p_in, float *p_out, int ndim,
                            int64_t nz, int64_t ny, int64_t nx, double wx,
                            double wy, double wz, int k, const double *sigma_k,
                            const double *hden_k, const double *tau_k,
                            const double *tl_k, const double *theta_k,
                            int flags, void *stream);
int nsol_pd_fusedk_iter_f64(const double *xbar_in, double *xbar_out,
                            const double *x_in, double *x_out,
                            const double *bt, const double *p_in,
                            double *p_out, int ndim, int64_t nz, int64_t ny,
                            int64_t nx, double wx, double wy, double wz, int k,
                            const double *sigma_k, const double *hden_k,
                            const double *tau_k, const double *tl_k,
                            const double *theta_k, int flags, void *stream);
/* Footprint shape and z-chunk of nsol_pd_fusedk_iter_* are tuned online for
 * volumes of >= 16 Mi voxels: the first few dozen launches of a new problem
 * shape each try one candidate (they are real launches of the run, timed with
 * events that are read back without synchronising), then the fastest is kept
 * for the life of the process.  Returns 1 once a shape has settled, 0 while it
 * is still exploring, -1 if the shape has not been seen. */
int nsol_pd_fusedk_tuned(int elem_size, int k, int64_t nz, int64_t ny, int64_t nx);
/* Number of k_pd_fusedk launches of depth k (2 or 3) this process has made
 * (any entry point, any shape); -1 for another k.  For tests and tools that
 * must know which kernel a run went through. */
int nsol_pd_fusedk_launches(int k);
/* 1 when nsol_pd_run_* sends a run's trailing PAIR of iterations through depth 2 of
 * k_pd_fusedk (large volume whose depth-3 plan has settled: the depth-2 form then takes
 * the same workgroup size and tile count without exploring), 0 when through
 * k_pd_fused2. */
int nsol_pd_fusedk_tail2(int elem_size, int64_t nz, int64_t ny, int64_t nx);
/* The settled configuration (waves per workgroup, tiles along x, z-chunk);
 * NSOL_EINVAL while the shape is unknown or still exploring. */
int nsol_pd_fusedk_plan(int elem_size, int k, int64_t nz, int64_t ny, int64_t nx,
                        int *waves, int *ntx, int64_t *zchunk);
/* The whole loop of primal_dual_solver.py:232-261 in ONE launch for cache-
 * resident volumes (BASELINE configs 1-2; at most one tile of <= 1024 lanes x one
 * 16-byte vector per CU, i.e. up to ~1 Mi float voxels): every workgroup keeps its
 * tile's x, xbar, b~, p in registers for all iterations and hands the one-voxel
 * faces to its face neighbours through tagged 8-byte granules in `ws`
 * (nsol_pdp.hip).  Bit-identical to `iterations` calls of nsol_pd_fused_iter_*.
 * nsol_pd_persist_run_*: xbar, x, p are updated in place (p is read only when
 * p_is_zero == 0).  nsol_pd_persist_run_to_*: the state is read from (xbar, x, p)
 * and the result written to (xbar_out, x_out, p_out) -- with distinct output
 * arrays the inputs survive a run that raised the error word, and the caller
 * repeats it through nsol_pd_run_* (one launch per iteration, the same bits).
 * The step sizes are host arrays as for nsol_pd_run_*.  ws: caller-owned DEVICE scratch
 * of nsol_pd_persist_ws_bytes() bytes, 16-byte aligned; a small set-up kernel
 * on `stream` fills it (the step sizes travel as its arguments).  err_word:
 * a 32-bit word the kernel can write, zeroed by the caller -- device memory or
 * pinned host memory the caller inspects after its own synchronisation --, or
 * NULL for the first word of ws (then cleared here).  It is non-zero after the
 * run if a workgroup gave up waiting for a neighbour (every wait is bounded;
 * the results are then invalid).
 * Returns -2 (nothing launched) when the kernel does not apply: rows not a
 * multiple of 16 bytes, unaligned arrays, more tiles than the device can hold
 * resident at once (occupancy of the kernel x CU count).
 * nsol_pd_persist_ws_bytes returns -1 in that case. */
int64_t nsol_pd_persist_ws_bytes(int elem_size, int ndim, int64_t nz, int64_t ny,
                                 int64_t nx, int iterations);
int nsol_pd_persist_run_f32(float *xbar, float *x, const float *bt, float *p, int ndim,
                            int64_t nz, int64_t ny, int64_t nx, double wx, double wy,
                            double wz, double lambda, const double *sigma_host,
                            const double *tau_host, const double *theta_host,
                            int iterations, int p_is_zero, double gamma_huber, int flags,
                            void *ws, int64_t ws_bytes, unsigned int *err_word,
                            void *stream);
int nsol_pd_persist_run_f64(double *xbar, double *x, const double *bt, double *p, int ndim,
                            int64_t nz, int64_t ny, int64_t nx, double wx, double wy,
                            double wz, double lambda, const double *sigma_host,
                            const double *tau_host, const double *theta_host,
                            int iterations, int p_is_zero, double gamma_huber, int flags,
                            void *ws, int64_t ws_bytes, unsigned int *err_word,
                            void *stream);
int nsol_pd_persist_run_to_f32(const float *xbar, const float *x, const float *bt,
                               const float *p, float *xbar_out, float *x_out,
                               float *p_out, int ndim, int64_t nz, int64_t ny, int64_t nx,
                               double wx, double wy, double wz, double lambda,
                               const double *sigma_host, const double *tau_host,
                               const double *theta_host, int iterations, int p_is_zero,
                               double gamma_huber, int flags, void *ws, int64_t ws_bytes,
                               unsigned int *err_word, void *stream);
int nsol_pd_persist_run_to_f64(const double *xbar, const double *x, const double *bt,
                               const double *p, double *xbar_out, double *x_out,
                               double *p_out, int ndim, int64_t nz, int64_t ny, int64_t nx,
                               double wx, double wy, double wz, double lambda,
                               const double *sigma_host, const double *tau_host,
                               const double *theta_host, int iterations, int p_is_zero,
                               double gamma_huber, int flags, void *ws, int64_t ws_bytes,
                               unsigned int *err_word, void *stream);
/* `iterations` iterations enqueued back to back with the host-side step
 * schedule (primal_dual_solver.py:278-403): sigma/tau/theta_host[n] are the
 * values used in iteration n.  xbar0/xbar1 and p0/p1 are ping-pong buffers;
 * every launch reads slot s and writes slot s^1, starting from slot 0.  Pairs
 * of iterations use the two-iteration kernel when x_alt (scratch of the size of
 * x, may be NULL) is given and the kernel applies.  On return x holds the final
 * primal iterate and *final_slot_host (may be NULL) the slot holding the final
 * xbar / p.  With NSOL_PD_RUN_X_MAY_SWAP in flags (and final_slot_host given) the
 * multi-iteration kernels' ping-pong of x is not undone by a copy: bit 1 of
 * *final_slot_host says whether the final primal iterate is in x_alt (2) or in x
 * (0), bit 0 is the slot of xbar / p.  p0 is treated as zero in iteration 0 when
 * p_is_zero != 0.  gamma_huber is the Huber parameter (0.05). */
int nsol_pd_run_f32(float *xbar0, float *xbar1, float *x, float *x_alt,
                    const float *bt, float *p0, float *p1, int ndim,
                    int64_t nz, int64_t ny, int64_t nx, double wx, double wy,
                    double wz, double lambda, const double *sigma_host,
                    const double *tau_host, const double *theta_host,
                    int iterations, int p_is_zero, double gamma_huber,
                    int flags, int *final_slot_host, void *stream);
int nsol_pd_run_f64(double *xbar0, double *xbar1, double *x, double *x_alt,
                    const double *bt, double *p0, double *p1, int ndim,
                    int64_t nz, int64_t ny, int64_t nx, double wx, double wy,
                    double wz, double lambda, const double *sigma_host,
                    const double *tau_host, const double *theta_host,
                    int iterations, int p_is_zero, double gamma_huber,
                    int flags, int *final_slot_host, void *stream);
/* The same run on a 3-D volume whose arrays hold their rows at a PITCH >= nx elements
 * (a multiple of 16 bytes; planes ny * pitch apart, the components of p nz * ny * pitch
 * apart): the layout for volumes whose rows are not whole 16-byte vectors (511^3,
 * 181 x 217 x 181 ...).  Accesses are aligned, a row's partial vector is masked in
 * registers exactly as in the contiguous ragged form (same bits in the valid elements),
 * and the padding behind a row's end may hold anything before and after the call (it
 * is never read as data).  x_alt as for nsol_pd_run_*; the single trailing iteration of a
 * run goes through nsol_pd_fused_iter_*'s kernel on the same strides.
 * nsol_pd_fusedk_iter_pitched_*: one launch of depth k = 2 / 3 on that layout;
 * nsol_pd_fusedk_tail2_pitched: see nsol_pd_fusedk_tail2. */
int nsol_pd_run_pitched_f32(float *xbar0, float *xbar1, float *x, float *x_alt,
                            const float *bt, float *p0, float *p1, int ndim, int64_t nz,
                            int64_t ny, int64_t nx, int64_t pitch, double wx, double wy,
                            double wz, double lambda, const double *sigma_host,
                            const double *tau_host, const double *theta_host,
                            int iterations, int p_is_zero, double gamma_huber, int flags,
                            int *final_slot_host, void *stream);
int nsol_pd_run_pitched_f64(double *xbar0, double *xbar1, double *x, double *x_alt,
                            const double *bt, double *p0, double *p1, int ndim, int64_t nz,
                            int64_t ny, int64_t nx, int64_t pitch, double wx, double wy,
                            double wz, double lambda, const double *sigma_host,
                            const double *tau_host, const double *theta_host,
                            int iterations, int p_is_zero, double gamma_huber, int flags,
                            int *final_slot_host, void *stream);
int nsol_pd_fusedk_iter_pitched_f32(const float *xbar_in, float *xbar_out, const float *x_in,
                                    float *x_out, const float *bt, const float *p_in,
                                    float *p_out, int ndim, int64_t nz, int64_t ny,
                                    int64_t nx, int64_t pitch, double wx, double wy,
                                    double wz, int k, const double *sigma,
                                    const double *hden, const double *tau, const double *tl,
                                    const double *theta, int flags, void *stream);
int nsol_pd_fusedk_iter_pitched_f64(const double *xbar_in, double *xbar_out,
                                    const double *x_in, double *x_out, const double *bt,
                                    const double *p_in, double *p_out, int ndim, int64_t nz,
                                    int64_t ny, int64_t nx, int64_t pitch, double wx,
                                    double wy, double wz, int k, const double *sigma,
                                    const double *hden, const double *tau, const double *tl,
                                    const double *theta, int flags, void *stream);
int nsol_pd_fusedk_tail2_pitched(int elem_size, int64_t nz, int64_t ny, int64_t nx,
                                 int64_t pitch);

/* ---------------------------------------------------------------------- *
 * ADMM outer update, admm_linear_solver.py:202-218, 239-253
 * ---------------------------------------------------------------------- */
/* t = grad(x) + w - c;  n = sqrt(sum_a t_a^2);  v_a = n > thr ?
 * max(n - thr, 0) * t_a / n : 0;  w = t - v;  rhs = rhs_scale * (v - w + c).
 * c (b_reg / x_scale) may be NULL (= 0).  w is updated in place; v and rhs may
 * each be NULL (the loop of admm_linear_solver.py:165-218 reads v only through the
 * next right-hand side: 12 of the 52 bytes per voxel less; its robust-loss
 * minimizers ignore b_reg -- tikhonov_linear_solver.py:201-208 -- so there the
 * right-hand side is not written either: 28 bytes).
 * grad is fused in (x is the primal volume). */
int nsol_admm_vw_update_f32(const float *x, float *v, float *w, const float *c,
                            float *rhs, int ndim, int64_t nz, int64_t ny,
                            int64_t nx, double wx, double wy, double wz,
                            double thr, double rhs_scale, void *stream);
int nsol_admm_vw_update_f64(const double *x, double *v, double *w,
                            const double *c, double *rhs, int ndim, int64_t nz,
                            int64_t ny, int64_t nx, double wx, double wy,
                            double wz, double thr, double rhs_scale,
                            void *stream);
/* The outer update AND the vector the next x-update's LSMR starts from, in one
 * pass (3-D volumes whose rows are whole 16-byte vectors; -2 otherwise, nothing
 * launched): what nsol_admm_vw_update_norm_* (v = NULL) followed by
 * nsol_lsmr_v_update_to_* (B = gradient, c_v = 0) leave --
 *   w_out = t - shrink(t) for t = grad(x) + w_in - c,
 *   g = c_atu * atb + c_btu * grad^T(rhs),  rhs = rhs_scale * (2 shrink(t) - t + c)
 * (admm_linear_solver.py:208-222 and the right-hand side A^T b + rho B^T(v - w + b_reg)
 * of tikhonov_linear_solver.py:146-158 on the normal equations) -- with rhs never
 * written: w_out and g are bit for bit those of the two kernels, result[0] = sum rhs^2,
 * result[1] = sum g^2 (doubles; ws: nsol_hip_reduce_ws_doubles() doubles).  w_in is
 * read at neighbouring voxels, so w_out must be another array. */
int nsol_admm_vw_update_g_f32(const float *x, const float *w_in, float *w_out,
                              const float *c, const float *atb, float *g, int ndim,
                              int64_t nz, int64_t ny, int64_t nx, double wx, double wy,
                              double wz, double thr, double rhs_scale, double c_atu,
                              double c_btu, double *result, double *ws, void *stream);
int nsol_admm_vw_update_g_f64(const double *x, const double *w_in, double *w_out,
                              const double *c, const double *atb, double *g, int ndim,
                              int64_t nz, int64_t ny, int64_t nx, double wx, double wy,
                              double wz, double thr, double rhs_scale, double c_atu,
                              double c_btu, double *result, double *ws, void *stream);
/* The same with *result = the sum of squares of the rhs written (double; ws:
 * nsol_hip_reduce_ws_doubles() doubles): with rhs_scale = sqrt(rho) the rhs is
 * the lower block of the right-hand side LSMR starts from
 * (tikhonov_linear_solver.py:232-236) and the sum its share of ||b||^2, so
 * neither a scaling pass nor a norm pass over it is needed.  rhs must not be
 * NULL. */
int nsol_admm_vw_update_norm_f32(const float *x, float *v, float *w, const float *c,
                                 float *rhs, int ndim, int64_t nz, int64_t ny,
                                 int64_t nx, double wx, double wy, double wz,
                                 double thr, double rhs_scale, double *result,
                                 double *ws, void *stream);
int nsol_admm_vw_update_norm_f64(const double *x, double *v, double *w,
                                 const double *c, double *rhs, int ndim, int64_t nz,
                                 int64_t ny, int64_t nx, double wx, double wy,
                                 double wz, double thr, double rhs_scale,
                                 double *result, double *ws, void *stream);
/* isotropic vector soft-threshold alone (admm_linear_solver.py:239-253):
 * t, v are ndim stacked blocks of m elements. */
int nsol_vector_shrink_f32(const float *t, float *v, int ndim, int64_t m,
                           double thr, void *stream);
int nsol_vector_shrink_f64(const double *t, double *v, int ndim, int64_t m,
                           double thr, void *stream);

/* ---------------------------------------------------------------------- *
 * Robust data term, linear_solver.py:315-340 + loss_functions.py:82-248
 * ---------------------------------------------------------------------- */
/* r = residual (A x - b).  g = rho'(r^2) * r (may alias r; may be NULL);
 * result[0] = 0.5 * sum rho(r^2).  loss = NSOL_LOSS_*; f_scale = data_loss_scale. */
int nsol_loss_cost_grad_f32(const float *r, float *g, int64_t n, int loss,
                            double f_scale, double *result, double *ws,
                            void *stream);
int nsol_loss_cost_grad_f64(const double *r, double *g, int64_t n, int loss,
                            double f_scale, double *result, double *ws,
                            void *stream);

/* The same with the residual formed on the way, r = ax - b (ax = A x; the
 * reference's `A(x) - b`, linear_solver.py:318): g may alias ax. */
int nsol_loss_residual_cost_grad_f32(const float *ax, const float *b, float *g,
                                     int64_t n, int loss, double f_scale,
                                     double *result, double *ws, void *stream);
int nsol_loss_residual_cost_grad_f64(const double *ax, const double *b, double *g,
                                     int64_t n, int loss, double f_scale,
                                     double *result, double *ws, void *stream);

/* element-wise rho(f2) and rho'(f2) (either output may be NULL); the API of
 * loss_functions.py:82-248 (huber's gamma is a parameter there, default 1.345). */
int nsol_loss_eval_f32(const float *f2, float *rho, float *drho, int64_t n,
                       int loss, double f_scale, double huber_gamma,
                       void *stream);
int nsol_loss_eval_f64(const double *f2, double *rho, double *drho, int64_t n,
                       int loss, double f_scale, double huber_gamma,
                       void *stream);
/* prior values of prior_measures.py:27-52 on a stacked field t (ndim blocks of
 * m): mode 0: result = sum_i sqrt(sum_a t_a[i]^2)          (total variation)
 *    mode 1: result = sum_i huber(sum_a t_a[i]^2; gamma) / (2 gamma)  (Huber) */
int nsol_vector_norm_sum_f32(const float *t, int ndim, int64_t m, int mode,
                             double gamma, double *result, double *ws,
                             void *stream);
int nsol_vector_norm_sum_f64(const double *t, int ndim, int64_t m, int mode,
                             double gamma, double *result, double *ws,
                             void *stream);

/* ---------------------------------------------------------------------- *
 * Fused LSMR vector kernels for [A; sqrt(alpha) B] (tikhonov_linear_solver.py
 * :226-274 driving scipy lsmr.py:320-413).  bmode: 0 = no regulariser block,
 * 1 = B is the gradient (ndim, extents, inverse spacings as above), 2 = B is the
 * identity.  Each call leaves the squared 2-norm of what it wrote in result[0].
 * ---------------------------------------------------------------------- */
#define NSOL_B_NONE 0
#define NSOL_B_GRAD 1
#define NSOL_B_IDENTITY 2
/* u_top = c_av*Av + c_u*u_top;  u_bot = c_bv*B(v) + c_u*u_bot.  Av may be NULL
 * when the top block has been updated by nsol_corr3_wrap_axpby_*: then only the
 * lower block is updated and summed (bmode must not be 0). */
int nsol_lsmr_u_update_f32(const float *Av, const float *v, float *u_top,
                           float *u_bot, int bmode, int ndim, int64_t nz,
                           int64_t ny, int64_t nx, double wx, double wy,
                           double wz, double c_av, double c_bv, double c_u,
                           double *result, double *ws, void *stream);
int nsol_lsmr_u_update_f64(const double *Av, const double *v, double *u_top,
                           double *u_bot, int bmode, int ndim, int64_t nz,
                           int64_t ny, int64_t nx, double wx, double wy,
                           double wz, double c_av, double c_bv, double c_u,
                           double *result, double *ws, void *stream);
/* v = c_atu*Atu + c_btu*B^T(u_bot) + c_v*v */
int nsol_lsmr_v_update_f32(const float *Atu, const float *u_bot, float *v,
                           int bmode, int ndim, int64_t nz, int64_t ny,
                           int64_t nx, double wx, double wy, double wz,
                           double c_atu, double c_btu, double c_v,
                           double *result, double *ws, void *stream);
int nsol_lsmr_v_update_f64(const double *Atu, const double *u_bot, double *v,
                           int bmode, int ndim, int64_t nz, int64_t ny,
                           int64_t nx, double wx, double wy, double wz,
                           double c_atu, double c_btu, double c_v,
                           double *result, double *ws, void *stream);
/* The same update written to v_out, which may be v (in place) or Atu (the new vector
 * takes the place of A^T u and the old v stays): for a caller that keeps every
 * Golub-Kahan vector v_k and assembles x = sum_k a_k v_k once at the end instead of
 * carrying h, hbar and x through every iteration (SciPy lsmr.py:352-364 -- 28 bytes
 * per element and iteration). */
int nsol_lsmr_v_update_to_f32(const float *Atu, const float *u_bot, const float *v,
                              float *v_out, int bmode, int ndim, int64_t nz, int64_t ny,
                              int64_t nx, double wx, double wy, double wz,
                              double c_atu, double c_btu, double c_v,
                              double *result, double *ws, void *stream);
int nsol_lsmr_v_update_to_f64(const double *Atu, const double *u_bot, const double *v,
                              double *v_out, int bmode, int ndim, int64_t nz, int64_t ny,
                              int64_t nx, double wx, double wy, double wz,
                              double c_atu, double c_btu, double c_v,
                              double *result, double *ws, void *stream);
/* hbar = h + c_hbar*hbar;  x = x + c_x*hbar;  h = c_v*v + c_h*h;
 * result[0] = sum x^2   (scipy lsmr.py:367-371, 407) */
int nsol_lsmr_hx_update_f32(float *hbar, float *x, float *h, const float *v,
                            int64_t n, double c_hbar, double c_x, double c_h,
                            double c_v, double *result, double *ws,
                            void *stream);
int nsol_lsmr_hx_update_f64(double *hbar, double *x, double *h,
                            const double *v, int64_t n, double c_hbar,
                            double c_x, double c_h, double c_v, double *result,
                            double *ws, void *stream);

/* The first-order Tikhonov regulariser's share of the robust-loss objective
 * (tikhonov_linear_solver.py:201-208, :229-236 with B = gradient, B_adj its
 * adjoint): grad = g + alpha * K^T(K x) and result[0] = sum |K x|^2 in one pass
 * over x instead of nsol_grad_*, nsol_dot_*, nsol_grad_adj_*, nsol_lincomb2_*
 * (grad is bit-identical to theirs).  g may alias grad; x may not.
 * ws: nsol_hip_reduce_ws_doubles() doubles. */
int nsol_tk1_reg_cost_grad_f32(const float *x, const float *g, float *grad, int ndim,
                               int64_t nz, int64_t ny, int64_t nx, double wx,
                               double wy, double wz, double alpha, double *result,
                               double *ws, void *stream);
int nsol_tk1_reg_cost_grad_f64(const double *x, const double *g, double *grad,
                               int ndim, int64_t nz, int64_t ny, int64_t nx,
                               double wx, double wy, double wz, double alpha,
                               double *result, double *ws, void *stream);
/* The same pass with what L-BFGS-B asks of every new gradient while it is in
 * registers (scipy's lnsrlb / projgr behind tikhonov_linear_solver.py:214-220):
 * result[0] = sum |K x|^2, result[1] = grad'd (d may be NULL: 0), result[2] =
 * max_i |P(x - grad)_i - x_i| for the bounds lo <= x <= hi (+-INFINITY: none), as
 * nsol_dot_* and nsol_lb_projgr_* return them; with gold (the gradient at the start of
 * the iteration; may be NULL) ydiff = grad - gold and result[3] = its sum of squares,
 * as nsol_lb_diff_dots_* (the y of the BFGS update, scipy's matupd).  grad as above.
 * ws: 4 * nsol_hip_reduce_ws_doubles() doubles; result: device double[4]. */
int nsol_tk1_reg_objective_f32(const float *x, const float *g, float *grad,
                               const float *d, const float *gold, float *ydiff,
                               int ndim, int64_t nz, int64_t ny,
                               int64_t nx, double wx, double wy, double wz,
                               double alpha, double lo, double hi, double *result,
                               double *ws, void *stream);
int nsol_tk1_reg_objective_f64(const double *x, const double *g, double *grad,
                               const double *d, const double *gold, double *ydiff,
                               int ndim, int64_t nz, int64_t ny,
                               int64_t nx, double wx, double wy, double wz,
                               double alpha, double lo, double hi, double *result,
                               double *ws, void *stream);
/* The same stencil for LSMR run as Lanczos on the normal equations
 * M = A^T A + alpha K^T K (lsmr_normal in nsol_amd/lsmr.py, behind
 * tikhonov_linear_solver.py:146-158):
 *   nsol_tk1_grad_norm_*: result[0] = sum |K x|^2 alone (x'K'K x of the Lanczos
 *     coefficient alfa; nothing but x is read, nothing written);
 *   nsol_tk1_lanczos_*:   out = c_g g + alpha K^T(K x) + c_x x + c_z z (z may be NULL)
 *     and result[0] = sum out^2 -- the three-term recurrence
 *     y_{j+1} = (A^T A y_j + alpha K'K y_j)/beta_j - (alfa/beta_j) y_j - (beta_j/beta_{j-1}) y_{j-1}
 *     and beta_{j+1}^2 in one pass.  g may alias out; x and z may not. */
int nsol_tk1_grad_norm_f32(const float *x, int ndim, int64_t nz, int64_t ny, int64_t nx,
                           double wx, double wy, double wz, double *result, double *ws,
                           void *stream);
int nsol_tk1_grad_norm_f64(const double *x, int ndim, int64_t nz, int64_t ny, int64_t nx,
                           double wx, double wy, double wz, double *result, double *ws,
                           void *stream);
int nsol_tk1_lanczos_f32(const float *x, const float *g, const float *z, float *out,
                         int ndim, int64_t nz, int64_t ny, int64_t nx, double wx,
                         double wy, double wz, double alpha, double c_g, double c_x,
                         double c_z, double *result, double *ws, void *stream);
int nsol_tk1_lanczos_f64(const double *x, const double *g, const double *z, double *out,
                         int ndim, int64_t nz, int64_t ny, int64_t nx, double wx,
                         double wy, double wz, double alpha, double c_g, double c_x,
                         double c_z, double *result, double *ws, void *stream);

/* ---------------------------------------------------------------------- *
 * Pair statistics for the evaluation measures of similarity_measures.py:26-120
 * (SSD, MAE, MSE, RMSE, PSNR, NCC).  result: device double[8] =
 *   { sum (x-mx)(y-my), sum (x-mx)^2, sum (y-my)^2, sum |x-y|, sum (x-y)^2,
 *     max y, sum x, sum y }.  ws: nsol_hip_reduce_ws_doubles() doubles.
 * ---------------------------------------------------------------------- */
int nsol_pair_stats_f32(const float *x, const float *y, int64_t n, double mx,
                        double my, double *result, double *ws, void *stream);
int nsol_pair_stats_f64(const double *x, const double *y, int64_t n, double mx,
                        double my, double *result, double *ws, void *stream);

/* ---------------------------------------------------------------------- *
 * Length-n pieces of a GPU-resident L-BFGS-B (nsol_amd/lbfgsb.py), replacing
 * the host loops of scipy.optimize.minimize(method="L-BFGS-B") behind
 * tikhonov_linear_solver.py:197-220.  Uniform bounds lo <= x <= hi (+-INFINITY
 * = absent).  iwhere: int8 per variable (0 free with bounds, -1 unbounded,
 * 1 / 2 fixed at lower / upper bound, -3 free with zero gradient, 3 fixed);
 * a NULL iwhere means "all variables".  result / ws as for the reductions.
 *   projgr        result[0] = max_i |projected gradient|
 *   mdot          result[0] = sum over free i of x[i]*y[i]
 *   cauchy_setup  classifies variables, d = -g on moving ones, tbk = breakpoint
 *                 (INFINITY if none); result[0..3] = sum d^2, #breakpoints,
 *                 #moving variables without a breakpoint and g != 0, #moving
 *   count_window  result[0] = number of breakpoints with (tbk, i) >
 *                 (t_done, i_done) and tbk <= t_hi
 *   select        the indices of those breakpoints, unordered; *count = how
 *                 many (may exceed capacity: then size the window first)
 *   cauchy_finish xcp = bound for breakpoints up to (t_done, i_done), else
 *                 x + tsum*d; updates iwhere
 *   wcomb         out = free ? scale*(sum_k bcoef[k]*base[k] + sum_j wcoef[j]*w[j]) : 0
 *                 (host arrays of device pointers; nbase <= 3, nw <= 40)
 *   project_step  xnew = free ? clip(xcp + d) : xcp; result[0] = #free at a bound
 *   ratio_min     result[0] = min over free i of the feasible step ratio,
 *                 result[1] = its smallest index (-1 if none)
 *   trunc_apply   xnew = free ? (i == ibd ? bound : xcp + alpha*d) : xcp
 * ---------------------------------------------------------------------- */
/* masked_gram: result[(i,j)], i <= j row by row, = sum over the free variables
 * (iwhere <= 0; all if iwhere is NULL) of vecs[i] * vecs[j], for nvec <= 24
 * vectors in ONE pass over them -- the Y'ZZ'Y, S'ZZ'S and S'ZZ'Y blocks of the
 * subspace matrix (2c^2 + c masked dots for c stored pairs).  vecs is a HOST
 * array of nvec device pointers (16-byte aligned); ws holds at least
 * nsol_lb_gram_ws_doubles() doubles. */
/* mdots: result[k] = sum over the free variables of vecs[k] * y, k < nvec, with y
 * and the mask read once (W^T v of the compact L-BFGS representation).  vecs is
 * a HOST array of device pointers; ws: nsol_lb_gram_ws_doubles() doubles. */
int nsol_lb_mdots_f32(const float *const *vecs, int nvec, const float *y,
                      const int8_t *iwhere, int64_t n, double *result, double *ws,
                      void *stream);
int nsol_lb_mdots_f64(const double *const *vecs, int nvec, const double *y,
                      const int8_t *iwhere, int64_t n, double *result, double *ws,
                      void *stream);
/* diff_dots: out = a - b, result[0] = sum out^2, result[1] = sum out * c (0 when c
 * is NULL) in one pass: the line search's d = z - x with d'd and g'd, and the BFGS
 * pair y = g - g_old with y'y (scipy lbfgsb.f lnsrlb / matupd behind
 * tikhonov_linear_solver.py:214-220).  ws: 4 * 1024 doubles. */
int nsol_lb_diff_dots_f32(const float *a, const float *b, const float *c, float *out,
                          int64_t n, double *result, double *ws, void *stream);
int nsol_lb_diff_dots_f64(const double *a, const double *b, const double *c,
                          double *out, int64_t n, double *result, double *ws,
                          void *stream);
/* The subspace step of an iteration in ONE pass over the stored vectors (scipy's
 * subsm after the solve with the subspace matrix, lnsrlb's d = z - x, matupd's new
 * row of S'S / S'Y; tikhonov_linear_solver.py:214-220):
 *   dsub = free ? scale * (r + sum_j wcoef[j] * w[j]) : 0      (nsol_lb_wcomb_*)
 *   xn   = free ? clip(xcp + dsub, lo, hi) : xcp               (nsol_lb_project_step_*)
 *   d    = xn - x                                              (nsol_lb_diff_dots_*)
 *   result = { #{free xn at a bound}, d'd, g'd, w[0]'d, ..., w[nw-1]'d,   (nsol_lb_mdots_*)
 *              -(smallest feasible step ratio along d from x) }        (nsol_lb_ratio_min_*)
 * with the arithmetic of those five, value for value.  w_host / wcoef_host: HOST
 * arrays of nw <= 24 device pointers / doubles; result: nw + 4 device doubles; ws:
 * nsol_lb_gram_ws_doubles() doubles.  Returns -2 (nothing launched) when n is not a
 * multiple of 16 bytes of elements or an array is not 16-byte aligned, or nw > 24. */
int nsol_lb_subspace_step_f32(const float *const *w_host, const double *wcoef_host, int nw,
                              const float *r, const float *xcp, const float *x,
                              const float *g, const int8_t *iwhere, int64_t n,
                              double scale, double lo, double hi, float *xn, float *d,
                              double *result, double *ws, void *stream);
int nsol_lb_subspace_step_f64(const double *const *w_host, const double *wcoef_host, int nw,
                              const double *r, const double *xcp, const double *x,
                              const double *g, const int8_t *iwhere, int64_t n,
                              double scale, double lo, double hi, double *xn, double *d,
                              double *result, double *ws, void *stream);
/* The same with r formed in the pass instead of read,
 *   r = free ? rb3[0]*xcp + rb3[1]*x + rb3[2]*g + sum_j rcoef[j]*w[j] : 0
 * (nsol_lb_wcomb_*'s sum term for term -- scipy's cmprlb), from the values the step
 * holds anyway: nsol_lb_masked_gram_rgrad_* then runs with r_out = NULL.  rb3_host /
 * rcoef_host: HOST arrays of 3 / nw doubles. */
int nsol_lb_subspace_step_r_f32(const float *const *w_host, const double *wcoef_host, int nw,
                                const double *rb3_host, const double *rcoef_host,
                                const float *xcp, const float *x, const float *g,
                                const int8_t *iwhere, int64_t n, double scale, double lo,
                                double hi, float *xn, float *d, double *result, double *ws,
                                void *stream);
int nsol_lb_subspace_step_r_f64(const double *const *w_host, const double *wcoef_host, int nw,
                                const double *rb3_host, const double *rcoef_host,
                                const double *xcp, const double *x, const double *g,
                                const int8_t *iwhere, int64_t n, double scale, double lo,
                                double hi, double *xn, double *d, double *result, double *ws,
                                void *stream);
int64_t nsol_lb_gram_ws_doubles(void);
int nsol_lb_masked_gram_f32(const float *const *vecs, int nvec, const int8_t *iwhere,
                            int64_t n, double *result, double *ws, void *stream);
int nsol_lb_masked_gram_f64(const double *const *vecs, int nvec, const int8_t *iwhere,
                            int64_t n, double *result, double *ws, void *stream);
/* The same pass also forming the reduced gradient of the subspace step
 * (scipy's cmprlb behind tikhonov_linear_solver.py:214-220):
 *   r_out = free ? bcoef3[0]*base3[0] + bcoef3[1]*base3[1] + bcoef3[2]*base3[2]
 *                  + sum_k wcoef[k]*vecs[k] : 0
 * -- nsol_lb_wcomb_*'s sum, term for term in the same order, without a pass of
 * its own over the nvec vectors.  base3 / bcoef3 / wcoef: HOST arrays (three
 * device pointers, three and nvec doubles).  result holds nvec more doubles behind
 * the nvec (nvec + 1) / 2 matrix entries: sum_free vecs[k] * b for
 * b = bcoef3[0]*base3[0] + bcoef3[1]*base3[1] + bcoef3[2]*base3[2] -- with the
 * matrix they give W'Z r = W'Z b + (W'Z W) wcoef on the host, which saves the
 * nsol_lb_mdots_* pass that forms the subspace right-hand side (nvec <= 20).
 * Returns -2 (nothing launched) where
 * the LDS-DMA staged kernel does not apply (n not a multiple of 16, unaligned
 * arrays, no room for the three extra rows): call nsol_lb_masked_gram_* and
 * nsol_lb_wcomb_* then.  r_out may be NULL (matrix and products only: the caller
 * leaves r to nsol_lb_subspace_step_r_*). */
int nsol_lb_masked_gram_rgrad_f32(const float *const *vecs, int nvec,
                                  const int8_t *iwhere, int64_t n, double *result,
                                  double *ws, const float *const *base3,
                                  const double *bcoef3, const double *wcoef,
                                  float *r_out, void *stream);
int nsol_lb_masked_gram_rgrad_f64(const double *const *vecs, int nvec,
                                  const int8_t *iwhere, int64_t n, double *result,
                                  double *ws, const double *const *base3,
                                  const double *bcoef3, const double *wcoef,
                                  double *r_out, void *stream);
int nsol_lb_count_free(const int8_t *iwhere, int64_t n, double *result,
                       double *ws, void *stream);
int nsol_lb_projgr_f32(const float *x, const float *g, int64_t n, double lo, double hi,
                       double *result, double *ws, void *stream);
int nsol_lb_mdot_f32(const float *x, const float *y, const int8_t *iwhere, int64_t n,
                     double *result, double *ws, void *stream);
int nsol_lb_cauchy_setup_f32(const float *x, const float *g, int64_t n, double lo,
                             double hi, int8_t *iwhere, float *d, float *tbk,
                             double *result, double *ws, void *stream);
int nsol_lb_select_f32(const float *tbk, int64_t n, double t_done, int64_t i_done,
                       double t_hi, int64_t *out_idx, int capacity, int *count,
                       void *stream);
int nsol_lb_count_window_f32(const float *tbk, int64_t n, double t_done,
                             int64_t i_done, double t_hi, double *result,
                             double *ws, void *stream);
int nsol_lb_gather_f32(const float *src, const int64_t *idx, int count, float *out,
                       void *stream);
int nsol_lb_cauchy_finish_f32(const float *x, const float *d, const float *tbk, int64_t n,
                              double lo, double hi, int8_t *iwhere, float *xcp,
                              double tsum, double t_done, int64_t i_done,
                              void *stream);
int nsol_lb_wcomb_f32(float *out, int64_t n, const int8_t *iwhere, double scale,
                      int nbase, const float *const *base_host,
                      const double *bcoef_host, int nw, const float *const *w_host,
                      const double *wcoef_host, void *stream);
/* out = clip(sum_k wcoef[k] * w[k], lo, hi): LSMR's solution assembled from its
 * stored vectors (nsol_amd/lsmr.py) and projected onto the solver's bounds
 * (tikhonov_linear_solver.py:142-143 applied to the result of :146-158) in one
 * pass -- nsol_lb_wcomb_*'s sum, term for term, then nsol_clip_*'s projection.
 * w / wcoef: HOST arrays of nw <= 40 device pointers / doubles. */
int nsol_lincomb_clip_f32(float *out, int64_t n, int nw, const float *const *w_host,
                          const double *wcoef_host, double lo, double hi,
                          void *stream);
int nsol_lb_project_step_f32(const float *xcp, const float *d, int64_t n, double lo,
                             double hi, const int8_t *iwhere, float *xnew,
                             double *result, double *ws, void *stream);
int nsol_lb_ratio_min_f32(const float *x, const float *d, int64_t n, double lo,
                          double hi, const int8_t *iwhere, double *result,
                          double *ws, void *stream);
int nsol_lb_trunc_apply_f32(const float *xcp, const float *d, int64_t n, double lo,
                            double hi, const int8_t *iwhere, double alpha,
                            int64_t ibd, float *xnew, void *stream);
int nsol_lb_projgr_f64(const double *x, const double *g, int64_t n, double lo, double hi,
                       double *result, double *ws, void *stream);
int nsol_lb_mdot_f64(const double *x, const double *y, const int8_t *iwhere, int64_t n,
                     double *result, double *ws, void *stream);
int nsol_lb_cauchy_setup_f64(const double *x, const double *g, int64_t n, double lo,
                             double hi, int8_t *iwhere, double *d, double *tbk,
                             double *result, double *ws, void *stream);
int nsol_lb_select_f64(const double *tbk, int64_t n, double t_done, int64_t i_done,
                       double t_hi, int64_t *out_idx, int capacity, int *count,
                       void *stream);
int nsol_lb_count_window_f64(const double *tbk, int64_t n, double t_done,
                             int64_t i_done, double t_hi, double *result,
                             double *ws, void *stream);
int nsol_lb_gather_f64(const double *src, const int64_t *idx, int count, double *out,
                       void *stream);
int nsol_lb_cauchy_finish_f64(const double *x, const double *d, const double *tbk, int64_t n,
                              double lo, double hi, int8_t *iwhere, double *xcp,
                              double tsum, double t_done, int64_t i_done,
                              void *stream);
int nsol_lb_wcomb_f64(double *out, int64_t n, const int8_t *iwhere, double scale,
                      int nbase, const double *const *base_host,
                      const double *bcoef_host, int nw, const double *const *w_host,
                      const double *wcoef_host, void *stream);
int nsol_lincomb_clip_f64(double *out, int64_t n, int nw, const double *const *w_host,
                          const double *wcoef_host, double lo, double hi,
                          void *stream);
int nsol_lb_project_step_f64(const double *xcp, const double *d, int64_t n, double lo,
                             double hi, const int8_t *iwhere, double *xnew,
                             double *result, double *ws, void *stream);
int nsol_lb_ratio_min_f64(const double *x, const double *d, int64_t n, double lo,
                          double hi, const int8_t *iwhere, double *result,
                          double *ws, void *stream);
int nsol_lb_trunc_apply_f64(const double *xcp, const double *d, int64_t n, double lo,
                            double hi, const int8_t *iwhere, double alpha,
                            int64_t ibd, double *xnew, void *stream);

/* Sort a compacted list of `count` variable indices by (tbk[index], index):
 * the order in which the Cauchy search meets its breakpoints.  tmp: device
 * scratch of nsol_lb_sort_tmp_bytes(count, sizeof element) bytes. */
int64_t nsol_lb_sort_tmp_bytes(int count, int elem_size);
int nsol_lb_sort_candidates_f32(const float *tbk, int64_t *idx, int count,
                                void *tmp, int64_t tmp_bytes, void *stream);
int nsol_lb_sort_candidates_f64(const double *tbk, int64_t *idx, int count,
                                void *tmp, int64_t tmp_bytes, void *stream);

/* Walk of `count` breakpoints sorted by nsol_lb_sort_candidates_* (generalized
 * Cauchy point) with prefix sums; see nsol_sort.hip.  params (device):
 * p[2col] | c[2col] | M[2col][2col] row-major (float64).  table:
 * nsol_lb_walk_table_doubles(count, col) doubles; tmp:
 * nsol_lb_walk_tmp_bytes(count) bytes; event: device int[2].  out (device
 * double[8 + 4*col]): [kdone, stopped (1) / not (0), index of a clamped f'' or
 * -1, t, variable index, f', f'', dt_min at breakpoint kdone-1, p[2col],
 * c[2col]]. */
int64_t nsol_lb_walk_table_doubles(int count, int col);
int64_t nsol_lb_walk_tmp_bytes(int count);
int nsol_lb_cauchy_walk_f32(const float *tbk, const float *d, const float *x,
                            const int64_t *idx, int count,
                            const float *const *wy_host,
                            const float *const *ws_host, int col, double theta,
                            double lo, double hi, double tj, double f1,
                            double f2, double f2_org, double dtm,
                            const double *params, double *table, void *tmp,
                            int64_t tmp_bytes, int *event, double *out,
                            void *stream);
int nsol_lb_cauchy_walk_f64(const double *tbk, const double *d, const double *x,
                            const int64_t *idx, int count,
                            const double *const *wy_host,
                            const double *const *ws_host, int col, double theta,
                            double lo, double hi, double tj, double f1,
                            double f2, double f2_org, double dtm,
                            const double *params, double *table, void *tmp,
                            int64_t tmp_bytes, int *event, double *out,
                            void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NSOL_HIP_H */
