// micro-benchmark: random 16-byte gathers from a small global table (L1/TA path) vs LDS, gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k_l1(const float4* __restrict__ tab, const unsigned* __restrict__ idx, float4* out, int iters, int mask) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned h = idx[t];
  float4 acc = make_float4(0, 0, 0, 0);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      h = h * 1664525u + 1013904223u;
      const float4 v = tab[(h >> 12) & mask];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  out[t] = acc;
}
__global__ void k_lds(const float4* __restrict__ tab, const unsigned* __restrict__ idx, float4* out, int iters, int mask) {
  extern __shared__ float4 lds[];
  for (int i = threadIdx.x; i <= mask; i += blockDim.x) lds[i] = tab[i];
  __syncthreads();
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned h = idx[t];
  float4 acc = make_float4(0, 0, 0, 0);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      h = h * 1664525u + 1013904223u;
      const float4 v = lds[(h >> 12) & mask];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  out[t] = acc;
}
int main() {
  const int threads = 1024, blocks = 256, iters = 2000;
  for (int entries : {256, 512, 1024, 2048, 4096}) {
    float4* tab; unsigned* idx; float4* out;
    hipMalloc(&tab, entries * 16); hipMalloc(&idx, threads * blocks * 4); hipMalloc(&out, threads * blocks * 16);
    hipMemset(tab, 0, entries * 16);
    unsigned* h = (unsigned*)malloc(threads * blocks * 4);
    for (int i = 0; i < threads * blocks; ++i) h[i] = rand();
    hipMemcpy(idx, h, threads * blocks * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which) {
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (which == 0) hipLaunchKernelGGL(k_l1, dim3(blocks), dim3(threads), 0, 0, tab, idx, out, iters, entries - 1);
        else hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(threads), entries * 16, 0, tab, idx, out, iters, entries - 1);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      }
      const double waveinstr = (double)threads * blocks / 64 * iters * 8;
      printf("%s table=%5d B : %.3f ms  %.1f ns/wave-gather/CU -> ~%.1f cycles@2.1GHz per wave-instr per CU\n", which ? "LDS" : "L1 ",
             entries * 16, ms, ms * 1e6 / (waveinstr / blocks), ms * 1e6 / (waveinstr / blocks) * 2.1);
    }
  }
  return 0;
}
