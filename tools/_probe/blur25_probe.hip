// How fast is ONE separable 25-tap pass (the taps of A'A for sigma = 2) with the one-pass
// blur's structure at 8 waves per workgroup (the z ring of 24 vectors needs 256 registers)?
// Against the 13-tap pass at 16 waves.  hipcc --offload-arch=gfx950 -O3 -std=c++17
// -ffp-contract=off -I include -I nsol_amd/csrc tools/_probe/blur25_probe.hip -o blur25_probe
#define NSOL_BLUR3_DMA_IMPL
#include "nsol_blur3_dma.hpp"
#include <cstdio>
#include <vector>
int nsol_blur3_zchunk = 0;
int nsol_blur3_dma_rag = 1;
namespace nsol { int g_dummy; }
using namespace nsol_blur3;

template <int NT, int NW>
float run(const float *x, float *out, int64_t n, int reps) {
  Taps<float> t;
  double s = 0;
  for (int i = 0; i < NT; ++i) { t.w[i] = (float)exp(-0.5 * (i - NT / 2) * (i - NT / 2) / 8.0); s += t.w[i]; }
  for (int i = 0; i < NT; ++i) t.w[i] /= (float)s;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) {
    int rc = launch_blur3_dma<float, 4, NT, NW, 0>(x, out, n, n, n, t, t, t, 0);
    if (rc) { printf("NT %d NW %d rc %d\n", NT, NW, rc); return -1; }
  }
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) launch_blur3_dma<float, 4, NT, NW, 0>(x, out, n, n, n, t, t, t, 0);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  const int64_t n = 512, N = n * n * n;
  float *x, *y;
  hipMalloc(&x, N * 4); hipMalloc(&y, N * 4);
  std::vector<float> h(N);
  for (int64_t i = 0; i < N; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.0f;
  hipMemcpy(x, h.data(), N * 4, hipMemcpyHostToDevice);
  printf("13 taps, 16 waves: %.4f ms\n", run<13, 16>(x, y, n, 300));
  printf("25 taps,  8 waves: %.4f ms\n", run<25, 8>(x, y, n, 200));
  printf("13 taps,  8 waves: %.4f ms\n", run<13, 8>(x, y, n, 300));
  printf("13 taps, 16 waves: %.4f ms\n", run<13, 16>(x, y, n, 300));
  printf("13 taps,  8 waves: %.4f ms\n", run<13, 8>(x, y, n, 300));
  return 0;
}
