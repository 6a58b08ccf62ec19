// Issue-rate microbenchmark of fp64 VALU chains (gfx950): how many INDEPENDENT dependency chains per wave does a SIMD need, at
// 1 / 2 waves per SIMD, to issue an fp64 instruction every 4 cycles?  (The MHD sweep runs 2 waves per SIMD at 256 VGPRs.)
// k<CH>: every lane runs CH independent FMA chains, ITER x 8 instructions in total per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
template <int CH>
__global__ void __launch_bounds__(256) k(double* out, double seed) {
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3 + i;
  const double b = seed * 0.999, c = seed * 1e-3;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int r = 0; r < 8 / CH; ++r)
#pragma unroll
      for (int i = 0; i < CH; ++i) a[i] = __builtin_fma(a[i], b, c);
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CH>
void run(double* d, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<CH><<<blocks, 256>>>(d, 1.37);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 4; ++r) k<CH><<<blocks, 256>>>(d, 1.37);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 4;
  const double insts_per_simd = (double)ITER * 8 * waves_per_simd;
  printf("v_fma_f64  chains/lane %d  waves/SIMD %d  %8.3f ms  %6.2f ns per wave-op per SIMD = %5.2f cycles at 2.4 GHz\n", CH, waves_per_simd, ms,
         ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
}
int main() {
  double* d; hipMalloc(&d, sizeof(double) * 256 * 256 * 8);
  for (int w : {1, 2, 3, 4}) { run<1>(d, w); run<2>(d, w); run<4>(d, w); run<8>(d, w); }
  return 0;
}
