// micro-benchmark of rq::samplesort_topk on its own: every workgroup selects + sorts the K smallest of
// cnt keys `reps` times, like the large-K finish of the scan but with nothing else on the CU.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/samplesort tools/micro/samplesort.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "../../rayuela.jl_amd/csrc/rq_topk.h"

__global__ __launch_bounds__(512, 4) void k_ss(const uint64_t *in, uint64_t *scratch, uint16_t *bkt, uint64_t *out,
                                               uint32_t cnt, uint32_t K, int reps, unsigned long long *stats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint64_t *src = in + (size_t)blockIdx.x * cnt;
  uint64_t *dst = scratch + (size_t)blockIdx.x * cnt;
  uint16_t *b = bkt + (size_t)blockIdx.x * cnt;
  uint64_t *o = out + (size_t)blockIdx.x * K;
  for (int r = 0; r < reps; ++r) {
    const unsigned long long t0 = (stats && threadIdx.x == 0) ? clock64() : 0;
    rq::samplesort_topk<512>(src, dst, b, cnt, K, smem, [o](uint32_t rank, uint64_t key) { o[rank] = key; }, stats);
    __syncthreads();
    if (stats && threadIdx.x == 0) atomicAdd(&stats[0], (unsigned long long)clock64() - t0);
  }
}

int main(int argc, char **argv) {
  const uint32_t K = argc > 1 ? atoi(argv[1]) : 10000;
  const uint32_t cnt = argc > 2 ? atoi(argv[2]) : 19000;
  const int reps = 20;
  const int lds_kb = argc > 3 ? atoi(argv[3]) : 72;      // 72: two workgroups per CU like inside the scan kernel; 41: three (a finish kernel of its own)
  for (int blocks : {256, 512, 768, 1024}) {
    const size_t nkeys = (size_t)blocks * cnt;
    uint64_t *in, *scr, *out;
    uint16_t *bkt;
    unsigned long long *stats;
    hipMalloc(&in, nkeys * 8); hipMalloc(&scr, nkeys * 8); hipMalloc(&bkt, nkeys * 2);
    hipMalloc(&out, (size_t)blocks * K * 8); hipMalloc(&stats, 16 * 8);
    std::vector<uint64_t> h(nkeys);
    for (size_t i = 0; i < nkeys; ++i) {
      // distances clustered like a distribution tail, ids unique
      const double u = (double)rand() / RAND_MAX;
      const uint32_t d = 0x40000000u + (uint32_t)(u * u * u * u * 4.0e6);
      h[i] = ((uint64_t)d << 32) | (uint32_t)i;
    }
    hipMemcpy(in, h.data(), nkeys * 8, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_ss), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int it = 0; it < 2; ++it) {
      hipMemset(stats, 0, 16 * 8);
      hipEventRecord(e0);
      // 72 KiB of LDS per workgroup, like the scan kernel: two workgroups per CU at 512 blocks
      hipLaunchKernelGGL(k_ss, dim3(blocks), dim3(512), lds_kb * 1024, 0, in, scr, bkt, out, cnt, K, reps, stats);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<uint64_t> ho((size_t)blocks * K);
    unsigned long long st[16];
    hipMemcpy(ho.data(), out, ho.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(st, stats, sizeof(st), hipMemcpyDeviceToHost);
    bool ok = true;
    for (int b = 0; b < blocks && ok; b += 37) {
      std::vector<uint64_t> ref(h.begin() + (size_t)b * cnt, h.begin() + (size_t)(b + 1) * cnt);
      std::sort(ref.begin(), ref.end());
      for (uint32_t i = 0; i < K; ++i) if (ref[i] != ho[(size_t)b * K + i]) { ok = false; break; }
    }
    const double tot = (double)st[0];
    printf("lds=%dKiB blocks=%d K=%u cnt=%u: %.1f us per select+sort (wall per WG) = %.0f sorts/ms  %s | sample-sort %.0f%% search %.0f%% scan+scatter %.0f%% rank %.0f%%\n",
           lds_kb, blocks, K, cnt, ms * 1e3 / reps, blocks * reps / ms, ok ? "ok" : "MISMATCH", 100 * st[9] / tot, 100 * st[10] / tot,
           100 * st[11] / tot, 100 * (tot - st[9] - st[10] - st[11]) / tot);
    hipFree(in); hipFree(scr); hipFree(bkt); hipFree(out); hipFree(stats);
  }
  return 0;
}
