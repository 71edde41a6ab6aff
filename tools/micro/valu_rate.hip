// Issue rate of single VALU instructions on gfx950 (cycles per wave64 instruction, 2 waves per SIMD all
// running the same dependent-free stream).  Used to pick the cheapest forms for the encode epilogue.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(x) x x x x x x x x
#define BODY(ASM)                                                        \
  for (int i = 0; i < iters; ++i) {                                      \
    REP8(asm volatile(ASM : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a), "v"(b));) \
  }
template <int K>
__global__ __launch_bounds__(512) void k(float *out, int iters) {
  if (K == 14 || K == 15) {
    double v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
    double a = out[0], b = out[1];
    if (K == 14) BODY("v_mov_b64 %0, %8\n v_mov_b64 %1, %8\n v_mov_b64 %2, %8\n v_mov_b64 %3, %8\n v_mov_b64 %4, %9\n v_mov_b64 %5, %9\n v_mov_b64 %6, %9\n v_mov_b64 %7, %9")
    if (K == 15) BODY("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9")
    out[blockIdx.x * 512 + threadIdx.x] = (float)(v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7);
    return;
  }
  float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
  float a = out[0], b = out[1];
  // every asm statement is 8 independent instructions
  if (K == 0) BODY("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9")
  if (K == 1) BODY("v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %8")
  if (K == 2) BODY("v_min3_f32 %0, %0, %8, %9\n v_min3_f32 %1, %1, %8, %9\n v_min3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9\n v_min3_f32 %4, %4, %8, %9\n v_min3_f32 %5, %5, %8, %9\n v_min3_f32 %6, %6, %8, %9\n v_min3_f32 %7, %7, %8, %9")
  if (K == 3) BODY("v_min_i32 %0, %0, %8\n v_min_i32 %1, %1, %8\n v_min_i32 %2, %2, %8\n v_min_i32 %3, %3, %8\n v_min_i32 %4, %4, %8\n v_min_i32 %5, %5, %8\n v_min_i32 %6, %6, %8\n v_min_i32 %7, %7, %8")
  if (K == 4) BODY("v_min3_i32 %0, %0, %8, %9\n v_min3_i32 %1, %1, %8, %9\n v_min3_i32 %2, %2, %8, %9\n v_min3_i32 %3, %3, %8, %9\n v_min3_i32 %4, %4, %8, %9\n v_min3_i32 %5, %5, %8, %9\n v_min3_i32 %6, %6, %8, %9\n v_min3_i32 %7, %7, %8, %9")
  if (K == 5) BODY("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8")
  if (K == 6) BODY("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8")
  if (K == 7) BODY("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %9\n v_mov_b32 %5, %9\n v_mov_b32 %6, %9\n v_mov_b32 %7, %9")
  if (K == 8) BODY("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n v_cmp_lt_f32 vcc, %4, %9\n v_cmp_lt_f32 vcc, %5, %9\n v_cmp_lt_f32 vcc, %6, %9\n v_cmp_lt_f32 vcc, %7, %9")
  if (K == 9) BODY("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %9, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_cndmask_b32 %6, %6, %9, vcc\n v_cndmask_b32 %7, %7, %9, vcc")
  if (K == 11) BODY("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cndmask_b32_e64 %4, %4, %9, s[20:21]\n v_cndmask_b32_e64 %5, %5, %9, s[20:21]\n v_cndmask_b32_e64 %6, %6, %9, s[20:21]\n v_cndmask_b32_e64 %7, %7, %9, s[20:21]")
  if (K == 12) BODY("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %8, vcc")
  if (K == 13) BODY("v_cmp_lt_f32 s[20:21], %0, %8\n v_cmp_lt_f32 s[22:23], %1, %8\n v_cmp_lt_f32 s[24:25], %2, %8\n v_cmp_lt_f32 s[26:27], %3, %8\n v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[22:23]\n v_cndmask_b32_e64 %2, %2, %8, s[24:25]\n v_cndmask_b32_e64 %3, %3, %8, s[26:27]")
  if (K == 10) BODY("v_min_u32 %0, %0, %8\n v_min_u32 %1, %1, %8\n v_min_u32 %2, %2, %8\n v_min_u32 %3, %3, %8\n v_min_u32 %4, %4, %8\n v_min_u32 %5, %5, %8\n v_min_u32 %6, %6, %8\n v_min_u32 %7, %7, %8")
  out[blockIdx.x * 512 + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}
int main() {
  float *out; hipMalloc(&out, 256 * 512 * 4); hipMemset(out, 0, 256 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  const char *names[16] = {"v_fma_f32", "v_min_f32", "v_min3_f32", "v_min_i32", "v_min3_i32", "v_max_f32", "v_add_f32", "v_mov_b32", "v_cmp_lt_f32", "v_cndmask vcc", "v_min_u32", "v_cndmask sgpr", "cmp+cndmask vcc x4", "4 cmp sgpr + 4 cndmask", "v_mov_b64", "v_pk_fma_f32"};
  for (int kk = 0; kk < 16; ++kk) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      switch (kk) {
#define C(N) case N: hipLaunchKernelGGL(k<N>, dim3(256), dim3(512), 0, 0, out, iters); break;
        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15)
      }
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    // per SIMD: 2 waves x iters x 64 instructions
    printf("%-24s %.2f cycles per wave64 instruction (2.4 GHz assumed)\n", names[kk], ms * 2.4e6 / (2.0 * iters * 64));
  }
  return 0;
}
