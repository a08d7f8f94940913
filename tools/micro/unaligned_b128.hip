// Microbenchmark for the plan of DESIGN section 8.0: are 16-byte buffer loads and
// stores whose addresses are only 4-byte aligned (rows of an odd number of
// floats) legal and how fast are they on gfx950?  Copies n floats with one
// buffer_load_dwordx4 / buffer_store_dwordx4 per lane from src + shift to
// dst + shift (shift in floats, 0..3), checks the result and prints GB/s.
//   hipcc --offload-arch=gfx950 -O3 -o unaligned_b128 unaligned_b128.hip
//   ./unaligned_b128 [n]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void k_copy(const float *src, float *dst, long nvec) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(src), 0, (unsigned)(nvec * 16), 0x00020000);
  const __amdgpu_buffer_rsrc_t rd =
      __builtin_amdgcn_make_buffer_rsrc(dst, 0, (unsigned)(nvec * 16), 0x00020000);
  for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < nvec;
       j += (long)gridDim.x * blockDim.x) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(j * 16), 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(v, rd, (unsigned)(j * 16), 0, 0);
  }
}

int main(int argc, char **argv) {
  const long n = argc > 1 ? atol(argv[1]) : (1L << 28);   // floats
  float *a, *b;
  hipMalloc(&a, (n + 8) * 4);
  hipMalloc(&b, (n + 8) * 4);
  std::vector<float> h(n + 8);
  for (long i = 0; i < n + 8; ++i) h[i] = (float)(i % 100003);
  hipMemcpy(a, h.data(), (n + 8) * 4, hipMemcpyHostToDevice);
  for (int shift = 0; shift < 4; ++shift) {
    hipMemset(b, 0, (n + 8) * 4);
    const long nvec = n / 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_copy<<<2048, 256>>>(a + shift, b + shift, nvec);
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) k_copy<<<2048, 256>>>(a + shift, b + shift, nvec);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> out(n + 8);
    hipMemcpy(out.data(), b, (n + 8) * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (long i = 0; i < nvec * 4; ++i) bad += out[shift + i] != h[shift + i];
    printf("{\"shift_floats\": %d, \"GBps\": %.0f, \"mismatches\": %ld, \"err\": \"%s\"}\n",
           shift, 2.0 * nvec * 16 / (ms / 10) / 1e6, bad,
           hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
