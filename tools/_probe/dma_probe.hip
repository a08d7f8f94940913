// Probes of global_load_lds behaviour on gfx950 (not part of the library):
//  1. 16-byte pieces from sources that are only 4-byte aligned
//  2. lanes switched off by EXEC leave their LDS slot untouched
//  3. 4-byte pieces (lane-linear 4-byte destination)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k_probe(const float *src, float *out, int shift, int mode) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 4 + 64];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 4 + 64; i += 64) lds[i] = -1.0f;
  __syncthreads();
  if (mode == 0) {          // unaligned 16-byte pieces
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void *)(src + shift + 4 * lane),
        (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
  } else if (mode == 1) {   // odd lanes off
    if (lane & 1)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void *)(src + shift + 4 * lane),
          (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
  } else {                  // 4-byte pieces, lanes 0..3 only, at an LDS offset
    if (lane < 4)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void *)(src + shift + 7 * lane),
          (__attribute__((address_space(3))) void *)(lds + 40), 4, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 64 * 4 + 64; i += 64) out[i] = lds[i];
}
int main() {
  const int n = 4096;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *d, *o;
  hipMalloc(&d, n * 4); hipMalloc(&o, 320 * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  std::vector<float> r(320);
  int bad = 0;
  for (int shift = 0; shift < 4; ++shift) {
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d, o, shift, 0);
    hipMemcpy(r.data(), o, 320 * 4, hipMemcpyDeviceToHost);
    int e = 0;
    for (int i = 0; i < 256; ++i) e += r[i] != (float)(shift + i);
    printf("mode0 shift %d: %d wrong (first %g %g %g %g %g)\n", shift, e, r[0], r[1], r[2], r[3], r[4]);
    bad += e;
  }
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d, o, 1, 1);
  hipMemcpy(r.data(), o, 320 * 4, hipMemcpyDeviceToHost);
  { int e = 0;
    for (int l = 0; l < 64; ++l) for (int k = 0; k < 4; ++k) {
      float want = (l & 1) ? (float)(1 + 4 * l + k) : -1.0f;
      e += r[4 * l + k] != want;
    }
    printf("mode1 (odd lanes only, shift 1): %d wrong (%g %g %g %g | %g %g %g %g)\n", e, r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
    bad += e; }
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d, o, 3, 2);
  hipMemcpy(r.data(), o, 320 * 4, hipMemcpyDeviceToHost);
  { int e = 0;
    for (int i = 0; i < 320; ++i) {
      float want = (i >= 40 && i < 44) ? (float)(3 + 7 * (i - 40)) : -1.0f;
      e += r[i] != want;
    }
    printf("mode2 (4-byte pieces, lanes 0-3): %d wrong (%g %g %g %g %g %g)\n", e, r[39], r[40], r[41], r[42], r[43], r[44]);
    bad += e; }
  printf(bad ? "PROBE_FAIL\n" : "PROBE_OK\n");
  return 0;
}
