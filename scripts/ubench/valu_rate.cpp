// Issue-rate microbenchmark of the fp64 VALU instructions the Riemann kernels are made of (gfx950).
// Each kernel runs ITER x 8 independent chains per lane; result: cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITER 2048
template <int OP>
__global__ void __launch_bounds__(256) k(double* out, double seed) {
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3 + i;
  const double b = seed * 0.999, c = seed * 1e-3;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) a[i] = __builtin_fma(a[i], b, c);
      if (OP == 1) a[i] = a[i] * b;
      if (OP == 2) a[i] = a[i] + c;
      if (OP == 3) a[i] = __builtin_amdgcn_rcp(a[i]);
      if (OP == 4) a[i] = __builtin_amdgcn_rsq(a[i]);
      if (OP == 5) a[i] = fmax(a[i], b);
      if (OP == 6) a[i] = (a[i] > b) ? c : a[i];                 // v_cmp + 2 v_cndmask
      if (OP == 7) a[i] = __builtin_amdgcn_ldexp(a[i], 1);
      if (OP == 8) a[i] = a[i] / b;                              // full IEEE division
      if (OP == 9) a[i] = sqrt(a[i] + 2.0);                      // add + full IEEE sqrt
      if (OP == 10) a[i] = __builtin_amdgcn_div_fixup(a[i], b, c);
    }
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
void run(const char* name, double* d, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<blocks, 256>>>(d, 1.37);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<OP><<<blocks, 256>>>(d, 1.37);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double insts_per_simd = (double)ITER * 8 * waves_per_simd;   // wave-instructions of the measured op per SIMD
  int clk; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  printf("%-28s waves/SIMD %d  %8.3f ms  %6.2f cycles per wave-op (at %d MHz)\n", name, waves_per_simd, ms, ms * 1e-3 * clk * 1e3 / insts_per_simd, clk / 1000);
}
int main() {
  double* d; hipMalloc(&d, sizeof(double) * 256 * 256 * 8);
  for (int w : {1, 2, 4}) {
    run<0>("v_fma_f64", d, w); run<1>("v_mul_f64", d, w); run<2>("v_add_f64", d, w);
    run<3>("v_rcp_f64", d, w); run<4>("v_rsq_f64", d, w); run<5>("v_max_f64", d, w);
    run<6>("cmp+cndmask x2", d, w); run<7>("v_ldexp_f64", d, w); run<8>("IEEE div (11 ops)", d, w);
    run<9>("add + IEEE sqrt (19 ops)", d, w); run<10>("v_div_fixup_f64", d, w);
  }
  return 0;
}
