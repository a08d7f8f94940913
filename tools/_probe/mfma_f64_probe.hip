// Throughput and C/D layout of v_mfma_f64_16x16x4_f64 on gfx950 (not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_rate(double *out, int iters) {
  d4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int i = 0; i < iters; ++i) {
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);
  }
  d4 s = acc0 + acc1 + acc2 + acc3;
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
__global__ void k_layout(const double *A, const double *B, double *D) {
  // A: 16 x 4 (row-major), B: 4 x 16; lane l supplies A[l & 15][l >> 4], B[l >> 4][l & 15]
  const int l = threadIdx.x;
  d4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}
int main() {
  double *out; hipMalloc(&out, 1024 * 256 * 8);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {256, 1024}) {          // 1 and 4 waves per SIMD
    hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(256), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)blocks * 4 / 1024.0 * iters * 4;   // MFMAs per SIMD
    printf("blocks %d: %.3f ms, %.1f ns per MFMA per SIMD (%.1f cycles at 2.4 GHz), %.1f TFLOP/s\n", blocks, ms,
           ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4, (double)blocks * 4 * iters * 4 * 2048 / (ms * 1e-3) / 1e12);
  }
  std::vector<double> A(64), B(64), D(256), R(256, 0.0);
  for (int i = 0; i < 64; ++i) { A[i] = 1 + i * 0.5; B[i] = 2 - i * 0.25; }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
  double *dA, *dB, *dD; hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
  hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 256; ++i) bad += D[i] != R[i];
  printf("layout (row = (lane >> 4) + 4 reg, col = lane & 15): %d of 256 wrong\n", bad);
  return 0;
}
