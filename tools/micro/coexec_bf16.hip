// How do v_mfma_f32_32x32x16_bf16 and ordinary VALU instructions share a SIMD on gfx950?
// One stream per wavefront: per iteration 4 MFMAs (CH = 0: four independent accumulators; CH = 1: two chains of 2; CH = 2: one
// dependent chain of 4), each followed by NV VALU instructions (v_min3_f32 / v_mov_b64 mix, independent registers).  Waves per
// SIMD = W (workgroup = 4 W waves).  Prints cycles per MFMA slot (iteration / 4) at the measured wall clock and a nominal 2.4 GHz.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NV, int CH, bool MFMA>
__global__ __launch_bounds__(1024) void k(float *out, int iters) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  bf16x8 x, y;
  for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(float)(threadIdx.x & 7); y[e] = (__bf16)1.0f; }
  float v[8];
  for (int r = 0; r < 8; ++r) v[r] = (float)threadIdx.x * 0.001f + r;
  const float p = out[0], q = out[1];
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (MFMA) {
        if (CH == 0) {
          if (s == 0) a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
          if (s == 1) a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
          if (s == 2) a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
          if (s == 3) a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
        } else if (CH == 1) {
          if (s & 1) a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
          else a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
        } else {
          a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < NV; ++r) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v[r & 7]) : "v"(p), "v"(q));
    }
  }
  float res = a0[0] + a1[1] + a2[2] + a3[3];
  for (int r = 0; r < 8; ++r) res += v[r];
  out[2 + blockIdx.x * 1024 + threadIdx.x] = res;
}

template <int NV, int CH, bool MFMA>
static void run(float *out, int W) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, CH, MFMA>), dim3(256), dim3(256 * W), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  }
  printf("W=%d NV=%2d chain=%d mfma=%d : %.3f ms  %.1f cycles per slot per SIMD @2.4GHz (slot = 1 MFMA + NV VALU of each of the W waves)\n",
         W, NV, CH, (int)MFMA, ms, ms * 2.4e6 / iters / 4);
}

int main() {
  float *out; hipMalloc(&out, (2 + 256 * 1024) * 4); hipMemset(out, 0, (2 + 256 * 1024) * 4);
  for (int W = 1; W <= 3; ++W) {
    run<0, 0, true>(out, W); run<4, 0, true>(out, W); run<6, 0, true>(out, W); run<8, 0, true>(out, W); run<12, 0, true>(out, W); run<16, 0, true>(out, W);
    run<8, 0, false>(out, W); run<16, 0, false>(out, W);
    run<8, 1, true>(out, W); run<8, 2, true>(out, W); run<16, 2, true>(out, W); run<0, 2, true>(out, W);
  }
  return 0;
}
