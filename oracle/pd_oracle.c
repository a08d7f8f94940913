/* CPU oracle, C restatement of the primal-dual denoising loop.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/nsol_oracle.py): loaded by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by nsol_amd.
 *
 * Same arithmetic, in the same order, as oracle/nsol_oracle.py's
 * primal_dual_denoise() -- which is pinned to the reference's golden vectors --
 * so the two agree bit for bit (tests/test_oracle_golden.py holds this file
 * to the goldens and to the NumPy restatement).  It exists because the NumPy
 * form needs 0.5 s per iteration at 128^3; this one runs the 500-iteration
 * depth of BASELINE configs 3 and 5 in seconds (OpenMP over planes / rows).
 *
 * Follows the reference (gift-surg/NSoL v0.1.14):
 *   loop            nsol/primal_dual_solver.py:232-261
 *   grad / grad_adj nsol/linear_operators.py:121-169 (zero-padded forward
 *                   differences [1,-1]/h and their exact transposes)
 *   prox_tv_conj    nsol/proximal_operators.py:138-140
 *   prox_huber_conj nsol/proximal_operators.py:156-159 (gamma = 0.05)
 *   prox_ell2 / ell1 denoising  nsol/proximal_operators.py:95-98, 117-120
 * Step sizes (primal_dual_solver.py:278-403) are passed in by the caller
 * (oracle/nsol_oracle.py: pd_schedule).
 *
 * Build: gcc -O2 -fopenmp -ffp-contract=off -fPIC -shared  (no -ffast-math:
 * IEEE double arithmetic without contraction, as NumPy does it).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static inline double clamp_unit(double q) {
  /* q / max(1, |q|) */
  double a = fabs(q);
  return q / (a > 1.0 ? a : 1.0);
}

/* One dual update for direction a: p_a = prox(p_a + sigma * (D_a xbar)).
 * (D u)[i] = u[i]*(-w) + u[i+1]*w, u := 0 past the last index. */
static void dual_axis(double *p, const double *xb, int64_t nz, int64_t ny,
                      int64_t nx, int64_t stride, int64_t len_axis, int axis,
                      double w, double sigma, int huber, double hden,
                      int first) {
  const int64_t n = nz * ny * nx;
  (void)n;
#pragma omp parallel for schedule(static)
  for (int64_t z = 0; z < nz; ++z)
    for (int64_t y = 0; y < ny; ++y) {
      const int64_t row = (z * ny + y) * nx;
      for (int64_t x = 0; x < nx; ++x) {
        const int64_t i = row + x;
        const int64_t pos = axis == 2 ? x : (axis == 1 ? y : z);
        double g = xb[i] * (-w);
        if (pos + 1 < len_axis) g += xb[i + stride] * w;
        double q = (first ? 0.0 : p[i]) + sigma * g;
        if (huber) q = q / hden;
        p[i] = clamp_unit(q);
      }
    }
}

/* Returns 0 on success.  b, x0, out: n = nz*ny*nx doubles (C order, x fastest);
 * ndim in {1,2,3} with the unused leading extents equal to 1; h[a] = spacing of
 * direction a (0 = x = last axis).  reg: 0 TV, 1 Huber.  data: 0 L2, 1 L1. */
int orc_pd_denoise(double *out, const double *b, const double *x0, int ndim,
                   int64_t nz, int64_t ny, int64_t nx, const double *h,
                   double x_scale, int reg, int data, double lambda,
                   const double *sig, const double *ta, const double *th,
                   int iterations) {
  const int64_t n = nz * ny * nx;
  double *x = (double *)malloc(sizeof(double) * n);
  double *xb = (double *)malloc(sizeof(double) * n);
  double *bt = (double *)malloc(sizeof(double) * n);
  double *p = (double *)malloc(sizeof(double) * n * ndim);
  if (!x || !xb || !bt || !p) {
    free(x); free(xb); free(bt); free(p);
    return 1;
  }
  const int64_t stride[3] = {1, nx, ny * nx};
  const int64_t len[3] = {nx, ny, nz};
  const int axis_of[3] = {2, 1, 0};
  double w[3] = {1.0, 1.0, 1.0};
  for (int a = 0; a < ndim; ++a) w[a] = 1.0 / h[a];
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    x[i] = x0[i] / x_scale;
    xb[i] = x[i];
    bt[i] = b[i] / x_scale;
  }
  for (int it = 0; it < iterations; ++it) {
    const double sigma = sig[it], tau = ta[it], theta = th[it];
    const double hden = 1. + sigma * 0.05;
    for (int a = 0; a < ndim; ++a)
      dual_axis(p + (int64_t)a * n, xb, nz, ny, nx, stride[a], len[a],
                axis_of[a], w[a], sigma, reg == 1, hden, it == 0);
    const double tl = tau * lambda;
#pragma omp parallel for schedule(static)
    for (int64_t z = 0; z < nz; ++z)
      for (int64_t y = 0; y < ny; ++y) {
        const int64_t row = (z * ny + y) * nx;
        for (int64_t xx = 0; xx < nx; ++xx) {
          const int64_t i = row + xx;
          const int64_t pos[3] = {xx, y, z};
          /* (D^T p)[i] = p[i]*(-w) + p[i-1]*w, p[-1] := 0; summed x, y, z */
          double ga = 0.0;
          for (int a = 0; a < ndim; ++a) {
            const double *pa = p + (int64_t)a * n;
            double t = pa[i] * (-w[a]);
            if (pos[a] > 0) t += pa[i - stride[a]] * w[a];
            ga = a == 0 ? t : ga + t;
          }
          const double u = x[i] - tau * ga;
          double xn;
          if (data == 0) {
            xn = (u + tl * bt[i]) / (1. + tl);
          } else {
            const double dlt = u - bt[i];
            double m = fabs(dlt) - tl;
            if (!(m > 0.0)) m = 0.0;
            const double sg = dlt > 0.0 ? 1.0 : (dlt < 0.0 ? -1.0 : 0.0);
            xn = bt[i] + m * sg;
          }
          xb[i] = xn + theta * (xn - x[i]);
          x[i] = xn;
        }
      }
  }
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) out[i] = x[i] * x_scale;
  free(x); free(xb); free(bt); free(p);
  return 0;
}
