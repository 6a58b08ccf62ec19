// Issue-rate microbenchmark, second pass (gfx950): long kernels (launch overhead < 1 %), inline-asm instruction streams.
//   * fp64 FMA / MUL / ADD / MAX alone, at 1..4 waves per SIMD;
//   * fp64 FMA interleaved 1:1 and 2:1 with 32-bit VALU work (v_add_u32, v_cndmask_b32, v_mov_b32) in the SAME wave:
//     does the 32-bit instruction cost an fp64 issue slot, or does it hide under the four passes of the fp64 one?
// The MHD sweep issues ~75 % fp64 arithmetic and ~25 % 32-bit integer / select instructions at 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 32768
#define F64(op) asm volatile(op " %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b))
#define FMA(i_) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(a[i_]) : "v"(a[i_]), "v"(b), "v"(c))
#define IADD(i_) asm volatile("v_add_u32 %0, %1, %2" : "=v"(x[i_]) : "v"(x[i_]), "v"(y))
#define CND(i_) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x[i_]) : "v"(x[i_]), "v"(y))
#define MOV(i_) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i_]) : "v"(y))
enum { K_FMA, K_MUL, K_ADD, K_MAX, K_IADD, K_FMA_IADD, K_FMA_CND, K_FMA_MOV, K_FMA2_IADD, K_FMA_IADD2, K_RCP, K_FMA4_RCP, K_NKINDS };
template <int KIND>
__global__ void __launch_bounds__(256) k(double* out, double seed, unsigned y) {
  double a[4];
  unsigned x[4];
  for (int i = 0; i < 4; ++i) { a[i] = seed + threadIdx.x * 1e-3 + i; x[i] = threadIdx.x + i; }
  const double b = seed * 0.999, c = seed * 1e-3;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (KIND == K_FMA) { FMA(i); FMA(i); }
      if (KIND == K_MUL) { F64("v_mul_f64"); F64("v_mul_f64"); }
      if (KIND == K_ADD) { F64("v_add_f64"); F64("v_add_f64"); }
      if (KIND == K_MAX) { F64("v_max_f64"); F64("v_max_f64"); }
      if (KIND == K_IADD) { IADD(i); IADD(i); }
      if (KIND == K_FMA_IADD) { FMA(i); IADD(i); FMA(i); IADD(i); }
      if (KIND == K_FMA_CND) { FMA(i); CND(i); FMA(i); CND(i); }
      if (KIND == K_FMA_MOV) { FMA(i); MOV(i); FMA(i); MOV(i); }
      if (KIND == K_FMA2_IADD) { FMA(i); FMA(i); IADD(i); }
      if (KIND == K_FMA_IADD2) { FMA(i); IADD(i); IADD(i); }
      if (KIND == K_RCP) { asm volatile("v_rcp_f64 %0, %1" : "=v"(a[i]) : "v"(a[i])); asm volatile("v_rcp_f64 %0, %1" : "=v"(a[i]) : "v"(a[i])); }
      if (KIND == K_FMA4_RCP) { FMA(i); FMA(i); FMA(i); FMA(i); asm volatile("v_rsq_f64 %0, %1" : "=v"(a[i]) : "v"(a[i])); }
    }
  }
  double s = 0;
  for (int i = 0; i < 4; ++i) s += a[i] + x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
static const char* names[] = {"fma64", "mul64", "add64", "max64", "add_u32", "fma64+add_u32 (1:1)", "fma64+cndmask (1:1)", "fma64+mov_b32 (1:1)",
                              "2 fma64 + 1 add_u32", "1 fma64 + 2 add_u32", "rcp64", "4 fma64 + 1 rsq64"};
static const int f64_per_iter[] = {8, 8, 8, 8, 0, 8, 8, 8, 8, 4, 8, 20}, all_per_iter[] = {8, 8, 8, 8, 8, 16, 16, 16, 12, 12, 8, 20};
template <int KIND>
void run(double* d, int w) {
  const int blocks = 256 * w;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<KIND><<<blocks, 256>>>(d, 1.37, 3u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<KIND><<<blocks, 256>>>(d, 1.37, 3u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double cyc = ms * 1e-3 * 2.4e9;
  const double n_all = (double)ITER * all_per_iter[KIND] * w, n_f64 = (double)ITER * f64_per_iter[KIND] * w;
  printf("%-24s waves/SIMD %d  %8.3f ms  %5.2f cycles per wave instruction", names[KIND], w, ms, cyc / n_all);
  if (n_f64 > 0 && n_f64 != n_all) printf("  = %5.2f per fp64 instruction", cyc / n_f64);
  printf("  (2.4 GHz)\n");
}
template <int KIND> void all(double* d) { for (int w : {1, 2, 3, 4}) run<KIND>(d, w); }
int main() {
  double* d; hipMalloc(&d, sizeof(double) * 256 * 256 * 8);
  all<K_FMA>(d); all<K_MUL>(d); all<K_ADD>(d); all<K_MAX>(d); all<K_IADD>(d); all<K_FMA_IADD>(d); all<K_FMA_CND>(d); all<K_FMA_MOV>(d);
  all<K_FMA2_IADD>(d); all<K_FMA_IADD2>(d); all<K_RCP>(d); all<K_FMA4_RCP>(d);
  return 0;
}
