// micro-benchmark of rq::bitonic_sort_tiled: nq sorts of p2 keys per workgroup, like the scan's final stage
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../rayuela.jl_amd/csrc/rq_topk.h"
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_sort(const uint64_t* in, uint64_t* out, uint32_t p2, uint32_t nconc, int reps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* scratch = reinterpret_cast<uint64_t*>(smem);
  const uint32_t tid = threadIdx.x, tps = THREADS / nconc, sg = tid / tps, sgi = tid % tps;
  uint64_t* a = scratch + (size_t)sg * p2;
  for (int r = 0; r < reps; ++r) {
    for (uint32_t i = sgi; i < p2; i += tps) a[i] = in[((size_t)blockIdx.x * nconc + sg) * p2 + i] + r;
    __syncthreads();
    rq::bitonic_sort_tiled(a, p2, tps / 64, sgi / 64, sgi & 63, true);
  }
  for (uint32_t i = sgi; i < p2; i += tps) out[((size_t)blockIdx.x * nconc + sg) * p2 + i] = a[i];
}
int main() {
  const int blocks = 512;
  for (int cfg = 0; cfg < 4; ++cfg) {
    const uint32_t p2 = cfg == 0 ? 1024 : cfg == 1 ? 1024 : cfg == 2 ? 4096 : 16384;
    const uint32_t nconc = cfg == 0 ? 8 : cfg == 1 ? 4 : cfg == 2 ? 2 : 1;
    const size_t nkeys = (size_t)blocks * nconc * p2;
    uint64_t *in, *out; hipMalloc(&in, nkeys * 8); hipMalloc(&out, nkeys * 8);
    uint64_t* h = (uint64_t*)malloc(nkeys * 8);
    for (size_t i = 0; i < nkeys; ++i) h[i] = ((uint64_t)rand() << 32) | (uint32_t)rand();
    hipMemcpy(in, h, nkeys * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 20;
    float ms = 0;
    for (int it = 0; it < 2; ++it) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_sort<512>, dim3(blocks), dim3(512), nconc * p2 * 8, 0, in, out, p2, nconc, reps);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    hipMemcpy(h, out, nkeys * 8, hipMemcpyDeviceToHost);
    bool ok = true;
    for (size_t b = 0; b < (size_t)blocks * nconc && ok; ++b)
      for (uint32_t i = 1; i < p2; ++i) if (h[b * p2 + i - 1] > h[b * p2 + i]) { ok = false; break; }
    // 2 workgroups per CU resident -> time per sort batch per WG
    printf("p2=%5u nconc=%u: %.3f ms total, %.1f us per batch of %u sorts (512 WGs on 256 CUs), sorted=%d\n", p2, nconc, ms,
           ms * 1e3 / reps, nconc, (int)ok);
    hipFree(in); hipFree(out); free(h);
  }
  return 0;
}
