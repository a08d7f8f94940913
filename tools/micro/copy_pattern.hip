// Microbenchmark: achieved bandwidth of the access pattern of k_pd_fusedk without
// its arithmetic -- a workgroup of NW*64 lanes laid out as rows x lxb vectors
// marches along z, reading six arrays and writing five (16 B per lane and array),
// with the same tile / z-chunk decomposition (halo lanes read but do not write).
//   hipcc --offload-arch=gfx950 -O3 -o copy_pattern copy_pattern.hip
//   ./copy_pattern n lxb rows halo_x halo_y zchunk nw [split_halo]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k_copy(const float *a0, const float *a1, const float *a2, const float *a3,
                       const float *a4, const float *a5, float *b0, float *b1, float *b2,
                       float *b3, float *b4, long nz, long ny, long nx, int lxb, int rows,
                       int hx, int hy, int ntx, int nty, int zchunk, int split_halo) {
  const int tid = threadIdx.x;
  int row = tid / lxb, lx = tid - row * lxb;
  if (split_halo) {
    // whole waves hold the 64 interior lanes of a row (aligned 1 KiB segments);
    // the two halo lanes of every row sit together in the last wave(s)
    const int main = rows * 64;
    if (tid < main) { row = tid >> 6; lx = 1 + (tid & 63); }
    else { const int h = tid - main; row = h >> 1; lx = (h & 1) ? 65 : 0; }
  }
  int bid = blockIdx.x;
  const int tx = bid % ntx; bid /= ntx;
  const int ty = bid % nty;
  const int zc = bid / nty;
  const long xv = lxb * 4 - 2 * hx, yv = rows - 2 * hy;
  const long x0 = (long)tx * xv - hx + lx * 4;
  const long y = (long)ty * yv - hy + row;
  const bool in = row < rows && x0 >= 0 && x0 < nx && y >= 0 && y < ny;
  const bool valid = in && x0 >= (long)tx * xv && x0 < (long)(tx + 1) * xv &&
                     y >= (long)ty * yv && y < (long)(ty + 1) * yv;
  long zb = (long)zc * zchunk, ze = zb + zchunk;
  if (ze > nz) ze = nz;
  for (long z = zb; z < ze; ++z) {
    const long o = (z * ny + y) * nx + x0;
    if (in) {
      f4 s = *(const f4 *)(a0 + o) + *(const f4 *)(a1 + o) + *(const f4 *)(a2 + o) +
             *(const f4 *)(a3 + o) + *(const f4 *)(a4 + o) + *(const f4 *)(a5 + o);
      if (valid) {
        *(f4 *)(b0 + o) = s; *(f4 *)(b1 + o) = s * 2.f; *(f4 *)(b2 + o) = s * 3.f;
        *(f4 *)(b3 + o) = s * 4.f; *(f4 *)(b4 + o) = s * 5.f;
      }
    }
  }
}

int main(int argc, char **argv) {
  const long n = argc > 1 ? atol(argv[1]) : 512;
  const int lxb = argc > 2 ? atoi(argv[2]) : 45, rows = argc > 3 ? atoi(argv[3]) : 17;
  const int hx = argc > 4 ? atoi(argv[4]) : 4, hy = argc > 5 ? atoi(argv[5]) : 2;
  const int zchunk = argc > 6 ? atoi(argv[6]) : 64, nw = argc > 7 ? atoi(argv[7]) : 12;
  const int split_halo = argc > 8 ? atoi(argv[8]) : 0;   // needs lxb = 66
  const long nv = n * n * n;
  std::vector<float *> in(6), out(5);
  for (auto &p : in) { hipMalloc(&p, nv * 4); hipMemset(p, 0, nv * 4); }
  for (auto &p : out) hipMalloc(&p, nv * 4);
  const long xv = lxb * 4 - 2 * hx, yv = rows - 2 * hy;
  const int ntx = (int)((n + xv - 1) / xv), nty = (int)((n + yv - 1) / yv);
  const int nzc = (int)((n + zchunk - 1) / zchunk);
  const long blocks = (long)ntx * nty * nzc;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0, 0);
    for (int i = 0; i < 5; ++i)
      hipLaunchKernelGGL(k_copy, dim3((unsigned)blocks), dim3(nw * 64), 0, 0, in[0], in[1],
                         in[2], in[3], in[4], in[5], out[0], out[1], out[2], out[3], out[4],
                         n, n, n, lxb, rows, hx, hy, ntx, nty, zchunk, split_halo);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms / 5 < best) best = ms / 5;
  }
  const double useful = 11.0 * nv * 4;
  const double lanes = (double)blocks * zchunk * rows * lxb * 16.0;
  printf("{\"n\": %ld, \"lxb\": %d, \"rows\": %d, \"halo\": [%d, %d], \"zchunk\": %d, "
         "\"waves\": %d, \"blocks\": %ld, \"ms\": %.4f, \"useful_GBps\": %.0f, "
         "\"requested_read_GB\": %.2f}\n",
         n, lxb, rows, hx, hy, zchunk, nw, blocks, best, useful / best / 1e6,
         lanes * 6 / 1e9);
  return 0;
}
