// k_blur3_dma<float, ...>: see nsol_blur3_dma.hpp
#define NSOL_BLUR3_DMA_IMPL
#include "nsol_blur3_dma.hpp"

namespace nsol_blur3 {
int blur3_dma_run(const float *x, float *out, int64_t nz, int64_t ny, int64_t nx,
                  const Taps<float> &tz, const Taps<float> &ty, const Taps<float> &tx, int ntaps,
                  int epi, double ca, double cb, double cc, double *result, double *part,
                  int64_t part_doubles, hipStream_t st) {
  return blur3_dma_dispatch<float>(x, out, nz, ny, nx, tz, ty, tx, ntaps, epi, ca, cb, cc,
                                result,
                                part, part_doubles, st);
}
int blur3_lanczos_a2(const float *y, float *t, int64_t nz, int64_t ny, int64_t nx,
                     const Taps<float> &tz, const Taps<float> &ty, const Taps<float> &tx, int ntaps,
                     double rho_g, double rho_i, double *board, int step, float *coef,
                     double *part, int64_t part_doubles, hipStream_t st) {
  const LanczosArgs<float> lz{nullptr, nullptr, nullptr, coef, board, step, rho_g, rho_i};
  return blur3_dma_dispatch<float>(y, t, nz, ny, nx, tz, ty, tx, ntaps, 2, 1.0, 1.0, 1.0,
                                nullptr, part, part_doubles, st, &lz);
}
}  // namespace nsol_blur3
