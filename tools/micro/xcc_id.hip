// What does s_getreg_b32 hwreg(HW_REG_XCC_ID) return per workgroup, and how does it relate to blockIdx % 8?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *o) {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) o[blockIdx.x] = x;
}
int main() {
  const int nb = 1024;
  unsigned *d, h[nb];
  hipMalloc(&d, nb * 4);
  hipLaunchKernelGGL(k, dim3(nb), dim3(512), 0, 0, d);
  hipMemcpy(h, d, nb * 4, hipMemcpyDeviceToHost);
  int same = 0;
  for (int i = 0; i < nb; ++i) same += ((h[i] & 7u) == (unsigned)(i & 7));
  printf("raw values of the first 16 blocks:");
  for (int i = 0; i < 16; ++i) printf(" 0x%x", h[i]);
  printf("\n(reg & 7) == blockIdx %% 8 for %d of %d blocks\n", same, nb);
  return 0;
}
