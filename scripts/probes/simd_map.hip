// probe: which SIMD does wave w of a workgroup run on (gfx950)?  HW_REG_HW_ID bits [5:4] = SIMD_ID
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}
int main() {
  unsigned *d, h[16 * 512];
  (void)hipMalloc(&d, sizeof h);
  for (int threads : {448, 576, 512, 320}) {
    (void)hipMemset(d, 0xff, sizeof h);
    k<<<512, threads, 100 * 1024>>>(d);
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("threads %d (waves %d):\n", threads, threads / 64);
    for (int b : {0, 1, 7, 100, 300, 511}) {
      printf("  block %3d simd:", b);
      for (int w = 0; w < threads / 64; ++w) printf(" %u", (h[b * 16 + w] >> 4) & 3);
      printf("   cu %u se %u\n", (h[b * 16] >> 8) & 15, (h[b * 16] >> 13) & 7);
    }
  }
  return 0;
}
