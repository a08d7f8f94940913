// Issue rates of v_cvt_f64_f32, v_fma_f64, v_fma_f32, v_cndmask on gfx950 (not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k_rate(float *out, int iters, float seed) {
  float f[8]; double d[8];
  for (int k = 0; k < 8; ++k) { f[k] = seed + threadIdx.x * 1e-3f + k; d[k] = f[k]; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (MODE == 0) { double t; asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(t) : "v"(f[k])); d[k] = t; }
      if (MODE == 1) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[k]) : "v"(d[(k + 1) & 7]), "v"(d[(k + 2) & 7]));
      if (MODE == 2) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[k]) : "v"(f[(k + 1) & 7]), "v"(f[(k + 2) & 7]));
      if (MODE == 3) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(f[k]) : "v"(f[(k + 1) & 7]), "v"(f[(k + 2) & 7]));
      if (MODE == 4) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(d[k]) : "v"(d[(k + 1) & 7]), "v"(d[(k + 2) & 7]));
      if (MODE == 5) asm volatile("v_add_f64 %0, %1, %2" : "=v"(d[k]) : "v"(d[(k + 1) & 7]), "v"(d[(k + 2) & 7]));
    }
  }
  float s = 0; for (int k = 0; k < 8; ++k) s += f[k] + (float)d[k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char *name, float *out) {
  const int iters = 20000, blocks = 1024;            // 4 waves per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = (double)blocks * 4 / 1024.0 * iters * 8;   // wave-instructions per SIMD
  printf("%-16s %.3f ms  %.2f ns per wave-instruction per SIMD (%.1f cycles at 2.4 GHz)\n", name, ms,
         ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
}
int main() {
  float *out; hipMalloc(&out, 1024 * 256 * 4);
  run<0>("v_cvt_f64_f32", out); run<1>("v_fma_f64", out); run<2>("v_fma_f32", out);
  run<3>("v_cndmask_b32", out); run<4>("v_mul_f64", out); run<5>("v_add_f64", out);
  return 0;
}
