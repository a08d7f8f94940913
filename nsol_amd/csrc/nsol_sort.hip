// Ordering of the breakpoints the L-BFGS-B Cauchy search has to walk
// (nsol_amd/lbfgsb.py): sorts a compacted list of variable indices by
// (breakpoint value, index).  The radix sorts are rocPRIM's (through hipCUB,
// a plain library primitive); the result is deterministic: indices are sorted
// first, then stably by key, so ties keep index order whatever order the
// compaction kernel produced.
#include <cstring>
#include <hipcub/hipcub.hpp>

#include "nsol_common.hpp"

using namespace nsol;

namespace {

template <typename T>
__global__ __launch_bounds__(kBlock) void k_keys(const T *__restrict__ tbk,
                                                  const int64_t *idx, int count,
                                                  T *keys) {
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < count;
       j += (int64_t)gridDim.x * blockDim.x)
    keys[j] = tbk[idx[j]];
}

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

template <typename T>
size_t cub_bytes(int count) {
  size_t a = 0, b = 0;
  // size queries: nothing is launched
  (void)hipcub::DeviceRadixSort::SortKeys(nullptr, a, (const int64_t *)nullptr,
                                          (int64_t *)nullptr, count);
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, (const T *)nullptr,
                                           (T *)nullptr, (const int64_t *)nullptr,
                                           (int64_t *)nullptr, count);
  return a > b ? a : b;
}

template <typename T>
int64_t total_bytes(int count) {
  return (int64_t)(align_up((size_t)count * 8) + 2 * align_up((size_t)count * sizeof(T)) +
                   align_up(cub_bytes<T>(count)) + 256);
}

template <typename T>
int sort_impl(const T *tbk, int64_t *idx, int count, void *tmp,
              int64_t tmp_bytes, void *stream) {
  if (count < 0 || !tbk || !idx || !tmp) return NSOL_EINVAL;
  if (count == 0) return 0;
  if (tmp_bytes < total_bytes<T>(count)) return NSOL_EINVAL;
  hipStream_t st = as_stream(stream);
  char *base = reinterpret_cast<char *>(
      (reinterpret_cast<uintptr_t>(tmp) + 255) & ~(uintptr_t)255);
  int64_t *idx_alt = reinterpret_cast<int64_t *>(base);
  base += align_up((size_t)count * 8);
  T *keys = reinterpret_cast<T *>(base);
  base += align_up((size_t)count * sizeof(T));
  T *keys_alt = reinterpret_cast<T *>(base);
  base += align_up((size_t)count * sizeof(T));
  size_t cb = cub_bytes<T>(count);
  hipError_t e = hipcub::DeviceRadixSort::SortKeys(base, cb, idx, idx_alt, count,
                                                   0, 64, st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(k_keys<T>, dim3(grid_for(count)), dim3(kBlock), 0, st, tbk,
                     idx_alt, count, keys);
  e = hipcub::DeviceRadixSort::SortPairs(base, cb, keys, keys_alt, idx_alt, idx,
                                         count, 0, (int)sizeof(T) * 8, st);
  if (e != hipSuccess) return (int)e;
  return launch_status();
}

// ---------------------------------------------------------------------------
// Walk of the sorted breakpoints (generalized Cauchy point, Byrd et al. 1995,
// algorithm CP) with prefix sums.  Between two stops the recurrences
//   p_k = p_{k-1} - g_k w_k,   c_k = c_{k-1} + dt_k p_{k-1},
//   f''_k = f''_{k-1} - theta g_k^2 + 2 g_k w_k'M p_{k-1} - g_k^2 w_k'M w_k,
//   f'_k  = f'_{k-1} + dt_k f''_{k-1} + g_k^2 - theta g_k z_k + g_k w_k'M c_k
// are linear, so all K candidate states come from 4*col + 2 inclusive scans
// (hipCUB) and three element-wise kernels; the first k with
// dt_min,k-1 < dt_k (the minimiser lies before breakpoint k) or with a clamped
// f'' ends the walk.  Table columns are float64 whatever the volume dtype.
// ---------------------------------------------------------------------------
constexpr int kMaxCol = 20;

template <typename T>
struct WalkPtrs {
  const T *wy[kMaxCol];
  const T *ws[kMaxCol];
};

struct WalkCols {          // column pointers into the caller's table
  double *t, *dt, *dib, *base1, *inc2, *f2cum, *inc1, *f1cum;
  double *W, *G, *Pcum, *Ccum;   // [2col][K]
};

inline WalkCols carve(double *table, int64_t K, int col) {
  WalkCols c;
  double *q = table;
  c.t = q; q += K; c.dt = q; q += K; c.dib = q; q += K; c.base1 = q; q += K;
  c.inc2 = q; q += K; c.f2cum = q; q += K; c.inc1 = q; q += K; c.f1cum = q; q += K;
  c.W = q; q += 2 * (int64_t)col * K;
  c.G = q; q += 2 * (int64_t)col * K;
  c.Pcum = q; q += 2 * (int64_t)col * K;
  c.Ccum = q;
  return c;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_walk_build(
    const T *__restrict__ tbk, const T *__restrict__ d, const T *__restrict__ x,
    const int64_t *__restrict__ idx, int K, WalkPtrs<T> P, int col, double theta,
    double lo, double hi, double tj, WalkCols C) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < K;
       k += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = idx[k];
    const double t = (double)tbk[i];
    const double tprev = k > 0 ? (double)tbk[idx[k - 1]] : tj;
    const double dib = (double)d[i];
    const double z = dib > 0 ? hi - (double)x[i] : lo - (double)x[i];
    C.t[k] = t;
    C.dt[k] = t - tprev;
    C.dib[k] = dib;
    C.base1[k] = dib * dib - theta * dib * z;
    C.inc2[k] = -theta * dib * dib;
    for (int j = 0; j < col; ++j) {
      const double wy = (double)P.wy[j][i];
      const double ws = theta * (double)P.ws[j][i];
      C.W[(int64_t)j * K + k] = wy;
      C.W[(int64_t)(col + j) * K + k] = ws;
      C.G[(int64_t)j * K + k] = dib * wy;
      C.G[(int64_t)(col + j) * K + k] = dib * ws;
    }
  }
}

// H_j[k] = dt_k * p_before_j[k],  p_before = p - Pcum[k-1]   (stored over G)
__global__ __launch_bounds__(kBlock) void k_walk_h(int K, int col2,
                                                    const double *params,
                                                    WalkCols C) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < K;
       k += (int64_t)gridDim.x * blockDim.x) {
    const double dt = C.dt[k];
    for (int j = 0; j < col2; ++j) {
      const double pb = params[j] - (k > 0 ? C.Pcum[(int64_t)j * K + k - 1] : 0.0);
      C.G[(int64_t)j * K + k] = dt * pb;
    }
  }
}

// second-order terms with M (params: p[2c] | c[2c] | M[2c][2c] row-major)
__global__ __launch_bounds__(kBlock) void k_walk_quad(int K, int col2,
                                                       const double *params,
                                                       WalkCols C) {
  const double *p0 = params, *c0 = params + col2, *M = params + 2 * col2;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < K;
       k += (int64_t)gridDim.x * blockDim.x) {
    double w[2 * kMaxCol];
    for (int j = 0; j < col2; ++j) w[j] = C.W[(int64_t)j * K + k];
    double wmc = 0.0, wmp = 0.0, wmw = 0.0;
    for (int i = 0; i < col2; ++i) {
      double v = 0.0;
      for (int j = 0; j < col2; ++j) v += M[i * col2 + j] * w[j];
      const double pb = p0[i] - (k > 0 ? C.Pcum[(int64_t)i * K + k - 1] : 0.0);
      const double ca = c0[i] + C.Ccum[(int64_t)i * K + k];
      wmc += ca * v;
      wmp += pb * v;
      wmw += w[i] * v;
    }
    const double dib = C.dib[k];
    C.inc2[k] += 2.0 * dib * wmp - dib * dib * wmw;
    C.base1[k] += dib * wmc;
  }
}

__global__ __launch_bounds__(kBlock) void k_walk_inc1(int K, double f2,
                                                       WalkCols C) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < K;
       k += (int64_t)gridDim.x * blockDim.x) {
    const double f2b = f2 + (k > 0 ? C.f2cum[k - 1] : 0.0);
    C.inc1[k] = C.dt[k] * f2b + C.base1[k];
  }
}

__global__ __launch_bounds__(kBlock) void k_walk_event(int K, double f1,
                                                        double f2, double f2_org,
                                                        double dtm, WalkCols C,
                                                        int *event) {
  const double eps = 2.220446049250313e-16;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < K;
       k += (int64_t)gridDim.x * blockDim.x) {
    const double f2a = f2 + C.f2cum[k];
    const double dtm_before =
        k > 0 ? -(f1 + C.f1cum[k - 1]) / (f2 + C.f2cum[k - 1]) : dtm;
    const bool stop = dtm_before < C.dt[k];
    const bool clamp = f2a < eps * f2_org;
    if (stop) atomicMin(&event[0], (int)k);
    if (clamp) atomicMin(&event[1], (int)k);
  }
}

// out: [kdone, stopped, clamp_at, t, idx, f1, f2, dtm, p_after[2c], c_after[2c]]
__global__ void k_walk_out(int K, int col2, double f1, double f2, double dtm,
                           const double *params, const int64_t *idx, WalkCols C,
                           const int *event, double *out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int kstop = event[0], kclamp = event[1];
  int kdone = K;
  double stopped = 0.0;
  if (kstop < kdone) { kdone = kstop; stopped = 1.0; }
  if (kclamp < kdone) { kdone = kclamp; stopped = 0.0; }
  out[0] = (double)kdone;
  out[1] = stopped;
  out[2] = (kclamp <= kdone && kclamp < K) ? (double)kclamp : -1.0;
  if (kdone > 0) {
    const int j = kdone - 1;
    const double f1a = f1 + C.f1cum[j], f2a = f2 + C.f2cum[j];
    out[3] = C.t[j];
    out[4] = (double)idx[j];
    out[5] = f1a;
    out[6] = f2a;
    out[7] = -f1a / f2a;
    for (int i = 0; i < col2; ++i) {
      out[8 + i] = params[i] - C.Pcum[(int64_t)i * K + j];
      out[8 + col2 + i] = params[col2 + i] + C.Ccum[(int64_t)i * K + j];
    }
  } else {
    out[3] = 0.0; out[4] = -1.0; out[5] = f1; out[6] = f2; out[7] = dtm;
  }
}

// The 2 col columns of a stage (G -> Pcum, then H -> Ccum) lie one behind the other in
// the table: ONE scan by key (key = position / K, the column) takes all of them -- the
// walk of a window then issues 4 scans whatever col is, where one scan per column made
// 4 col + 2 (84 tiny launches per window at ten stored pairs).
struct ColumnOf {
  int64_t K;
  __host__ __device__ int operator()(int64_t pos) const { return (int)(pos / K); }
};
typedef hipcub::TransformInputIterator<int, ColumnOf, hipcub::CountingInputIterator<int64_t>>
    ColumnKeys;

int g_walk_by_key = 1;           // 0: one hipCUB scan per column (the A/B form)

inline size_t scan_bytes(int K) {
  size_t b = 0;
  (void)hipcub::DeviceScan::InclusiveSum(nullptr, b, (const double *)nullptr,
                                         (double *)nullptr, K);
  // (the scan by key over all columns of a stage at the largest memory)
  size_t bk = 0;
  const ColumnKeys keys(hipcub::CountingInputIterator<int64_t>(0), ColumnOf{K > 0 ? K : 1});
  (void)hipcub::DeviceScan::InclusiveSumByKey(nullptr, bk, keys, (const double *)nullptr,
                                              (double *)nullptr,
                                              (size_t)2 * kMaxCol * (size_t)(K > 0 ? K : 1));
  return b > bk ? b : bk;
}

template <typename T>
int walk_impl(const T *tbk, const T *d, const T *x, const int64_t *idx, int K,
              const T *const *wy_host, const T *const *ws_host, int col,
              double theta, double lo, double hi, double tj, double f1,
              double f2, double f2_org, double dtm, const double *params,
              double *table, void *tmp, int64_t tmp_bytes, int *event,
              double *out, void *stream) {
  if (K < 1 || col < 0 || col > kMaxCol || !tbk || !d || !x || !idx || !table ||
      !tmp || !event || !out || (col > 0 && (!wy_host || !ws_host || !params)))
    return NSOL_EINVAL;
  const size_t sb = scan_bytes(K);
  if ((size_t)tmp_bytes < sb) return NSOL_EINVAL;
  hipStream_t st = as_stream(stream);
  WalkPtrs<T> P;
  for (int j = 0; j < kMaxCol; ++j) {
    P.wy[j] = j < col ? wy_host[j] : nullptr;
    P.ws[j] = j < col ? ws_host[j] : nullptr;
  }
  const WalkCols C = carve(table, K, col);
  const int col2 = 2 * col;
  const dim3 g(grid_for(K)), b(kBlock);
  size_t tb = (size_t)tmp_bytes;
  hipLaunchKernelGGL(k_walk_build<T>, g, b, 0, st, tbk, d, x, idx, K, P, col,
                     theta, lo, hi, tj, C);
  hipError_t ce = hipSuccess;            // first failure of a hipCUB scan
  auto scan = [&](const double *in, double *outp) {
    const hipError_t r = hipcub::DeviceScan::InclusiveSum(tmp, tb, in, outp, K, st);
    if (ce == hipSuccess) ce = r;
  };
  auto scan_columns = [&](const double *in, double *outp) {
    if (g_walk_by_key && col2 > 1) {
      const ColumnKeys keys(hipcub::CountingInputIterator<int64_t>(0), ColumnOf{K});
      const hipError_t r = hipcub::DeviceScan::InclusiveSumByKey(
          tmp, tb, keys, in, outp, (size_t)col2 * (size_t)K, hipcub::Equality(), st);
      if (ce == hipSuccess) ce = r;
    } else {
      for (int j = 0; j < col2; ++j) scan(in + (int64_t)j * K, outp + (int64_t)j * K);
    }
  };
  scan_columns(C.G, C.Pcum);
  if (col2 > 0) {
    hipLaunchKernelGGL(k_walk_h, g, b, 0, st, K, col2, params, C);
    scan_columns(C.G, C.Ccum);
    hipLaunchKernelGGL(k_walk_quad, g, b, 0, st, K, col2, params, C);
  }
  scan(C.inc2, C.f2cum);
  hipLaunchKernelGGL(k_walk_inc1, g, b, 0, st, K, f2, C);
  scan(C.inc1, C.f1cum);
  if (ce != hipSuccess) return (int)ce;
  const int init[2] = {0x7fffffff, 0x7fffffff};
  hipError_t e = hipMemcpyAsync(event, init, sizeof(init), hipMemcpyHostToDevice, st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(k_walk_event, g, b, 0, st, K, f1, f2, f2_org, dtm, C, event);
  hipLaunchKernelGGL(k_walk_out, dim3(1), dim3(64), 0, st, K, col2, f1, f2, dtm,
                     params, idx, C, event, out);
  return launch_status();
}

}  // namespace

extern "C" {
/* experiment knob of this file: "sort_walk_by_key" */
int nsol_hip_set_param_sort(const char *name, int value) {
  if (!name || strcmp(name, "sort_walk_by_key")) return NSOL_EINVAL;
  g_walk_by_key = value;
  return 0;
}
int64_t nsol_lb_walk_table_doubles(int count, int col) {
  return (int64_t)(8 + 8 * (int64_t)col) * (count > 0 ? count : 1);
}
int64_t nsol_lb_walk_tmp_bytes(int count) {
  return (int64_t)scan_bytes(count > 0 ? count : 1) + 256;
}
int nsol_lb_cauchy_walk_f32(const float *tbk, const float *d, const float *x,
                            const int64_t *idx, int count,
                            const float *const *wy_host,
                            const float *const *ws_host, int col, double theta,
                            double lo, double hi, double tj, double f1,
                            double f2, double f2_org, double dtm,
                            const double *params, double *table, void *tmp,
                            int64_t tmp_bytes, int *event, double *out,
                            void *stream) {
  return walk_impl<float>(tbk, d, x, idx, count, wy_host, ws_host, col, theta, lo,
                          hi, tj, f1, f2, f2_org, dtm, params, table, tmp,
                          tmp_bytes, event, out, stream);
}
int nsol_lb_cauchy_walk_f64(const double *tbk, const double *d, const double *x,
                            const int64_t *idx, int count,
                            const double *const *wy_host,
                            const double *const *ws_host, int col, double theta,
                            double lo, double hi, double tj, double f1,
                            double f2, double f2_org, double dtm,
                            const double *params, double *table, void *tmp,
                            int64_t tmp_bytes, int *event, double *out,
                            void *stream) {
  return walk_impl<double>(tbk, d, x, idx, count, wy_host, ws_host, col, theta,
                           lo, hi, tj, f1, f2, f2_org, dtm, params, table, tmp,
                           tmp_bytes, event, out, stream);
}
int64_t nsol_lb_sort_tmp_bytes(int count, int elem_size) {
  if (count <= 0) return 256;
  return elem_size == 4 ? total_bytes<float>(count) : total_bytes<double>(count);
}
int nsol_lb_sort_candidates_f32(const float *tbk, int64_t *idx, int count,
                                void *tmp, int64_t tmp_bytes, void *stream) {
  return sort_impl<float>(tbk, idx, count, tmp, tmp_bytes, stream);
}
int nsol_lb_sort_candidates_f64(const double *tbk, int64_t *idx, int count,
                                void *tmp, int64_t tmp_bytes, void *stream) {
  return sort_impl<double>(tbk, idx, count, tmp, tmp_bytes, stream);
}
}
