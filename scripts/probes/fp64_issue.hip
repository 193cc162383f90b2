// probe: issue cost of fp64 VALU instructions on gfx950 with 1, 2, 3, 4 waves per SIMD (one block per CU)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int MODE>
__global__ void k(double *out, int iters, double a, double b) {
  double x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = a + threadIdx.x * 1e-9 + i;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) x[i] = x[i] * b;                 // v_mul_f64
        if (MODE == 1) x[i] = x[i] + b;                 // v_add_f64
        if (MODE == 2) x[i] = __builtin_fma(x[i], b, a);  // v_fma_f64
        if (MODE == 3) x[i] = __builtin_fmax(x[i], b + i);  // v_max_f64
        if (MODE == 4) { float f = (float)x[i]; f = f * (float)b + 1.0f; x[i] = f; }  // cvt+fma f32
        if (MODE == 5) x[i] = (x[i] > b) ? x[i] : a;    // cmp + cndmask x2
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + (double)(t1 - t0) * 1e-30;
}
template <int MODE> void run(const char *name, int threads) {
  double *out; hipMalloc(&out, 256 * 1024 * 8);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<256, threads>>>(out, 100, 1.0, 1.0000001);
  hipEventRecord(e0);
  k<MODE><<<256, threads>>>(out, iters, 1.0, 1.0000001);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_wave = (double)iters * 32;
  const int wps = (threads / 64 + 3) / 4;
  printf("%-10s threads %4d (max %d waves/SIMD): %.3f ms, %.2f ns per instr per wave, %.2f ns per instr per SIMD (x2.4 GHz = %.2f cyc)\n",
         name, threads, wps, ms, ms * 1e6 / instr_per_wave, ms * 1e6 / instr_per_wave / wps, ms * 1e6 / instr_per_wave / wps * 2.4);
  hipFree(out);
}
int main() {
  for (int th : {64, 256, 448, 512, 768, 1024}) {
    run<0>("mul_f64", th); run<1>("add_f64", th); run<2>("fma_f64", th); run<3>("max_f64", th); run<5>("cmp+sel", th);
  }
  return 0;
}
