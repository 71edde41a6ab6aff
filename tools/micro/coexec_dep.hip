// Does the overlap of v_mfma_f32_32x32x16_bf16 with VALU work survive when the VALU work READS MFMA results (as the encode filter's
// min tree does)?  Per iteration: 3 chained MFMAs on accumulator A (then B, alternating), and a min3 tree + 8 v_mov_b64-style copies
// over the OTHER accumulator (finished one iteration earlier).  mode 0: VALU reads the accumulators; mode 1: the same instruction
// mix on plain registers (no dependence on the MFMAs).  W waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float tree(const f32x16 &a) {
  float m1 = __builtin_fminf(__builtin_fminf(a[0], a[1]), a[2]);
  float m2 = __builtin_fminf(__builtin_fminf(a[3], a[4]), a[5]);
  float m3 = __builtin_fminf(__builtin_fminf(a[6], a[7]), a[8]);
  float m4 = __builtin_fminf(__builtin_fminf(a[9], a[10]), a[11]);
  float m5 = __builtin_fminf(__builtin_fminf(a[12], a[13]), a[14]);
  float mm = __builtin_fminf(__builtin_fminf(m1, m2), m3);
  mm = __builtin_fminf(__builtin_fminf(mm, m4), m5);
  return __builtin_fminf(mm, a[15]);
}

template <int MODE, int EXTRA>
__global__ __launch_bounds__(1024) void k(float *out, int iters) {
  f32x16 a = {0}, b = {0}, keep = {0}, plain;
  for (int r = 0; r < 16; ++r) plain[r] = (float)threadIdx.x + r;
  bf16x8 x, y;
  for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(float)(threadIdx.x & 7); y[e] = (__bf16)1.0f; }
  float best = 1e30f;
  float v[8];
  for (int r = 0; r < 8; ++r) v[r] = (float)threadIdx.x * 0.001f + r;
  const float p = out[0], q = out[1];
  for (int i = 0; i < iters; ++i) {
    // tile t + 1 on A while B (finished) is reduced, then the other way round
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x16 &cur = half ? b : a;
      const f32x16 &done = half ? a : b;
      cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, cur, 0, 0, 0);
      cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, cur, 0, 0, 0);
      cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, cur, 0, 0, 0);
      const f32x16 &src = MODE == 0 ? done : plain;
      const float mm = tree(src);
      if (mm < best) { best = mm; keep = src; }
      if (MODE == 1) plain[i & 15] += 1.0f;
#pragma unroll
      for (int r = 0; r < EXTRA; ++r) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v[r & 7]) : "v"(p), "v"(q));
    }
  }
  float res = best + keep[3] + a[0] + b[1];
  for (int r = 0; r < 8; ++r) res += v[r];
  out[2 + blockIdx.x * 1024 + threadIdx.x] = res;
}

template <int MODE, int EXTRA>
static void run(float *out, int W) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 10000;
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, EXTRA>), dim3(256), dim3(256 * W), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  }
  printf("W=%d mode=%d (%s) extra=%2d : %.3f ms  %.1f nominal cycles per tile per SIMD (3 MFMA + tree + copy%s per wave)\n", W, MODE,
         MODE == 0 ? "VALU reads MFMA results" : "VALU on plain registers", EXTRA, ms, ms * 2.4e6 / iters / 2, EXTRA ? " + extra min3" : "");
}

int main() {
  float *out; hipMalloc(&out, (2 + 256 * 1024) * 4); hipMemset(out, 0, (2 + 256 * 1024) * 4);
  for (int W = 1; W <= 3; ++W) {
    run<0, 0>(out, W); run<1, 0>(out, W); run<0, 16>(out, W); run<1, 16>(out, W);
  }
  return 0;
}
