// Ordering of the breakpoints the L-BFGS-B Cauchy search has to walk
// (nsol_amd/lbfgsb.py): sorts a compacted list of variable indices by
// (breakpoint value, index).  The radix sorts are rocPRIM's (through hipCUB,
// a plain library primitive); the result is deterministic: indices are sorted
// first, then stably by key, so ties keep index order whatever order the
// compaction kernel produced.
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
  hipcub::DeviceRadixSort::SortKeys(nullptr, a, (const int64_t *)nullptr,
                                    (int64_t *)nullptr, count);
  hipcub::DeviceRadixSort::SortPairs(nullptr, b, (const T *)nullptr,
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

}  // namespace

extern "C" {
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
