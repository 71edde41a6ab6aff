// Do f32 MFMA and f32 VALU from DIFFERENT wavefronts of one SIMD overlap on gfx950?
// Workgroup = 8 waves (2 per SIMD).  mode 0: all waves MFMA; 1: all waves VALU; 2: half the waves MFMA, half VALU
// (pairing 0: waves 0-3 MFMA, 4-7 VALU; pairing 1: even / odd) -- whichever pairing puts one of each on a SIMD
// shows whether the two pipes run side by side.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int VK>
__global__ __launch_bounds__(512) void k(float *out, int mode, int iters, int pairing) {
  const int wave = threadIdx.x >> 6;
  bool do_mfma, do_valu;
  if (mode == 0) { do_mfma = true; do_valu = false; }
  else if (mode == 1) { do_mfma = false; do_valu = true; }
  else {
    const bool first = pairing == 0 ? (wave < 4) : ((wave & 1) == 0);
    do_mfma = first; do_valu = !first;
  }
  float res = 0;
  if (do_mfma) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    const float x = (float)threadIdx.x, y = 1.0f;
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    }
    res = a0[0] + a1[1] + a2[2] + a3[3];
  }
  if (do_valu) {
    if (VK == 0) {          // packed f32 FMA
      f32x2 v[8];
      for (int r = 0; r < 8; ++r) v[r] = f32x2{(float)threadIdx.x * 0.001f + r, 1.0f};
      const f32x2 m = {0.999f, 1.001f}, c = {0.5f, 0.25f};
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] = __builtin_elementwise_fma(v[r], m, c);
      }
      for (int r = 0; r < 8; ++r) res += v[r].x + v[r].y;
    } else if (VK == 1) {   // scalar f32 FMA
      float v[8];
      for (int r = 0; r < 8; ++r) v[r] = (float)threadIdx.x * 0.001f + r;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] = __builtin_fmaf(v[r], 0.999f, 0.5f);
      }
      for (int r = 0; r < 8; ++r) res += v[r];
    } else if (VK == 2) {   // f32 min / max
      float v[8];
      for (int r = 0; r < 8; ++r) v[r] = (float)threadIdx.x * 0.001f + r;
      float lim = out[0];
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
          for (int r = 0; r < 8; ++r) { asm volatile("v_min_f32 %0, %0, %1" : "+v"(v[r]) : "v"(lim)); }
      }
      for (int r = 0; r < 8; ++r) res += v[r];
    } else if (VK == 3) {   // integer add
      unsigned v[8];
      for (int r = 0; r < 8; ++r) v[r] = threadIdx.x + r;
      unsigned inc = (unsigned)iters;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
          for (int r = 0; r < 8; ++r) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[r]) : "v"(inc)); }
      }
      for (int r = 0; r < 8; ++r) res += (float)v[r];
    } else {                // v_cndmask / v_mov
      float v[8];
      for (int r = 0; r < 8; ++r) v[r] = (float)threadIdx.x * 0.001f + r;
      float alt = out[0];
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
          for (int r = 0; r < 8; ++r) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[r]) : "v"(alt) : ); }
      }
      for (int r = 0; r < 8; ++r) res += v[r];
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = res;
}

int main() {
  float *out; hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  const char *names[5] = {"v_pk_fma_f32", "v_fma_f32", "v_min_f32", "v_add_u32", "v_cndmask_b32"};
  for (int vk = 0; vk < 5; ++vk)
    for (int mode = 0; mode < 3; ++mode) {
      if (mode == 0 && vk > 0) continue;
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        switch (vk) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, out, mode, iters, 0); break;
          case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, out, mode, iters, 0); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, out, mode, iters, 0); break;
          case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, out, mode, iters, 0); break;
          default: hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, out, mode, iters, 0); break;
        }
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      }
      printf("%-14s mode %d (%s): %.3f ms  (%.0f cycles/iter)\n", names[vk], mode,
             mode == 0 ? "8 waves MFMA" : mode == 1 ? "8 waves VALU" : "4 MFMA + 4 VALU", ms, ms * 2.4e6 / iters);
    }
  return 0;
}
