// Throughput and operand / result layout of v_mfma_f64_4x4x4_4b_f64 on gfx950 (not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k_rate(double *out, int iters) {
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[k], 0, 0, 0);
  }
  double s = 0; for (int k = 0; k < 8; ++k) s += acc[k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// with a v_cvt_f64_f32 + v_cndmask pair per MFMA beside it (the Gram kernel's mix)
__global__ __launch_bounds__(256) void k_rate_mix(double *out, const float *in, int iters) {
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float f = in[threadIdx.x];
  double a = threadIdx.x * 1e-3;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      double t; asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(t) : "v"(f));
      acc[k] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, t, acc[k], 0, 0, 0);
    }
  }
  double s = 0; for (int k = 0; k < 8; ++k) s += acc[k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ void k_one(const double *a, const double *b, double *d) {
  const int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
int main() {
  double *out; hipMalloc(&out, 1024 * 256 * 8);
  float *in; hipMalloc(&in, 1024); hipMemset(in, 0, 1024);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mix = 0; mix < 2; ++mix)
  for (int blocks : {256, 1024}) {          // 1 and 4 waves per SIMD
    if (mix) hipLaunchKernelGGL(k_rate_mix, dim3(blocks), dim3(256), 0, 0, out, in, 10);
    else hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(256), 0, 0, out, 10);
    hipEventRecord(e0);
    if (mix) hipLaunchKernelGGL(k_rate_mix, dim3(blocks), dim3(256), 0, 0, out, in, iters);
    else hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)blocks * 4 / 1024.0 * iters * 8;   // MFMAs per SIMD
    printf("%s blocks %d: %.3f ms, %.2f ns per MFMA per SIMD (%.1f cycles at 2.4 GHz), %.1f TFLOP/s\n",
           mix ? "mfma+cvt" : "mfma", blocks, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4,
           (double)blocks * 4 * iters * 8 * 512 / (ms * 1e-3) / 1e12);
  }
  // layout: lane l supplies one element of A and one of B; try the natural maps
  std::vector<double> A(64), B(64), D(64);
  double *dA, *dB, *dD; hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 512);
  // the three 2-bit fields of the lane number, in every order, for A and for B
  const int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
  auto field = [](int l, int f) { return f == 0 ? (l & 3) : f == 1 ? ((l >> 2) & 3) : (l >> 4); };
  double a[4][4][4], b[4][4][4];               // [block][i][k], [block][k][j]
  for (int q = 0; q < 4; ++q) for (int i = 0; i < 4; ++i) for (int k = 0; k < 4; ++k) {
    a[q][i][k] = 1 + q * 16 + i * 4 + k + 0.5 * (i == k);
    b[q][k][i] = 3 - 0.25 * (q * 16 + k * 4 + i) + (q == i);
  }
  for (int pa = 0; pa < 6; ++pa) for (int pb = 0; pb < 6; ++pb) {
    // fields (blk, idx, k) of A and of B
    for (int l = 0; l < 64; ++l) {
      A[l] = a[field(l, perms[pa][0])][field(l, perms[pa][1])][field(l, perms[pa][2])];
      B[l] = b[field(l, perms[pb][0])][field(l, perms[pb][2])][field(l, perms[pb][1])];
    }
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 512, hipMemcpyDeviceToHost);
    int found = 0; char map[64][32];
    for (int l = 0; l < 64; ++l) {
      map[l][0] = 0;
      for (int q = 0; q < 4; ++q) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        double r = 0; for (int k = 0; k < 4; ++k) r += a[q][i][k] * b[q][k][j];
        if (r == D[l] && !map[l][0]) { snprintf(map[l], 32, "(%d,%d,%d)", q, i, j); ++found; }
      }
    }
    if (found >= 32) printf("A fields (blk,idx,k)=(%d,%d,%d) B fields=(%d,%d,%d): %d of 64 lanes hold an entry (block,i,j)\n",
           perms[pa][0], perms[pa][1], perms[pa][2], perms[pb][0], perms[pb][1], perms[pb][2], found);
    if (found == 64) { for (int l = 0; l < 64; ++l) printf("%s%s", map[l], (l & 15) == 15 ? "\n" : " "); }
  }
  return 0;
}
