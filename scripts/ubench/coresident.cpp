// Can a light streaming kernel share the CUs with a resident heavy kernel (158 KB LDS, 2 waves per SIMD of ~216 VGPRs)?  (gfx950)
//   heavy<V>: 512-thread workgroups, static LDS 158 KB, V VGPRs (amdgpu_num_vgpr + a live array), spins on fp64 FMAs for `iters` rounds
//   light<W>: 256-thread workgroups, launch_bounds(256, W) (W waves per SIMD -> 512 / W VGPRs), grid-stride copy of `n` doubles
// Each alone, then both on two streams: if they co-reside the pair takes ~max, otherwise ~sum.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// the register allocation of a kernel is the highest register it names: the asm clobber pins it
#define HEAVY(NAME, TOPREG, LDSKB)                                                                  \
__global__ void __launch_bounds__(512) NAME(double* out, int iters, double seed) {                 \
  asm volatile("v_mov_b32 " TOPREG ", 0" ::: TOPREG);                                               \
  heavy_body<LDSKB>(out, iters, seed);                                                              \
}
#define LIGHT(NAME, TOPREG, PRIO)                                                                   \
__global__ void __launch_bounds__(256) NAME(const double* __restrict__ src, double* __restrict__ dst, size_t n) { \
  asm volatile("v_mov_b32 " TOPREG ", 0" ::: TOPREG);                                               \
  __builtin_amdgcn_s_setprio(PRIO);                                                                 \
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) dst[i] = src[i] * 1.0000001; \
}
template <int LDSKB>
__device__ __forceinline__ void heavy_body(double* out, int iters, double seed) {
  constexpr int NV = 24;
  __shared__ double lds[LDSKB * 1024 / 8];
  double a[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) a[i] = seed + i + threadIdx.x * 1e-3;
  lds[threadIdx.x] = seed;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NV; ++i) a[i] = __builtin_fma(a[i], 0.999999, a[(i + 1) % NV] * 1e-9);
  }
  double s = lds[(threadIdx.x * 7) & 511];
#pragma unroll
  for (int i = 0; i < NV; ++i) s += a[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

HEAVY(heavy216, "v215", 158) HEAVY(heavy192, "v191", 158) HEAVY(heavy128, "v127", 158)
HEAVY(heavy216_lds120, "v215", 120) HEAVY(heavy216_lds64, "v215", 64) HEAVY(heavy216_lds8, "v215", 8)
LIGHT(light64, "v63", 0) LIGHT(light80, "v79", 0) LIGHT(light96, "v95", 0) LIGHT(light128, "v127", 0)
LIGHT(light64p, "v63", 3) LIGHT(light80p, "v79", 3) LIGHT(light48p, "v47", 3) LIGHT(light32p, "v31", 3)

// an update-like stream: 26 input arrays, 8 output arrays (34 x 8 B per element), one persistent 256-thread workgroup per CU
#define STREAM(NAME, TOPREG, PRIO, NIN)                                                             \
__global__ void __launch_bounds__(256) NAME(const double* __restrict__ src, double* __restrict__ dst, size_t n) { \
  asm volatile("v_mov_b32 " TOPREG ", 0" ::: TOPREG);                                               \
  __builtin_amdgcn_s_setprio(PRIO);                                                                 \
  const size_t m = n / 34;                                                                          \
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < m; i += (size_t)gridDim.x * 256ull) {      \
    double v[NIN];                                                                                  \
    _Pragma("unroll") for (int a = 0; a < NIN; ++a) v[a] = src[i + a * m];                          \
    double acc = 0;                                                                                 \
    _Pragma("unroll") for (int a = 0; a < NIN; ++a) acc = __builtin_fma(v[a], 1.0000001, acc);      \
    _Pragma("unroll") for (int a = 0; a < 8; ++a) __builtin_nontemporal_store(acc + a, &dst[i + a * m]); \
  }                                                                                                 \
}
STREAM(stream80p, "v79", 3, 26) STREAM(stream80, "v79", 0, 26) STREAM(stream64p, "v63", 3, 26)
typedef void (*heavy_t)(double*, int, double);
typedef void (*light_t)(const double*, double*, size_t);
void run(heavy_t H, light_t L, double* out, double* src, double* dst, size_t n, int iters) {
  hipStream_t s1, s2;
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1, f0, f1;
  hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&f0); hipEventCreate(&f1);
  hipFuncAttributes fa, fb;
  CHECK(hipFuncGetAttributes(&fa, (const void*)H)); CHECK(hipFuncGetAttributes(&fb, (const void*)L));
  float th, tl, tp_h, tp_l;
  hipLaunchKernelGGL(H, 256, 512, 0, s1, out, 10, 1.0); hipLaunchKernelGGL(L, 4096, 256, 0, s2, src, dst, n); CHECK(hipDeviceSynchronize());
  hipEventRecord(e0, s1); hipLaunchKernelGGL(H, 256, 512, 0, s1, out, iters * 4, 1.0); hipEventRecord(e1, s1); CHECK(hipDeviceSynchronize()); hipEventElapsedTime(&th, e0, e1);
  hipEventRecord(f0, s2); hipLaunchKernelGGL(L, 4096, 256, 0, s2, src, dst, n); hipEventRecord(f1, s2); CHECK(hipDeviceSynchronize()); hipEventElapsedTime(&tl, f0, f1);
  hipEventRecord(e0, s1); hipLaunchKernelGGL(H, 256, 512, 0, s1, out, iters * 4, 1.0); hipEventRecord(e1, s1);
  hipEventRecord(f0, s2); hipLaunchKernelGGL(L, 4096, 256, 0, s2, src, dst, n); hipEventRecord(f1, s2);
  CHECK(hipDeviceSynchronize());
  hipEventElapsedTime(&tp_h, e0, e1); hipEventElapsedTime(&tp_l, f0, f1);
  float tot; hipEventElapsedTime(&tot, e0, f1);
  printf("heavy %3d VGPRs (LDS %zu)  light %3d VGPRs: alone %.2f / %.2f ms   together: heavy %.2f  light %.2f  (start of heavy to end of light %.2f)\n",
         fa.numRegs, (size_t)fa.sharedSizeBytes, fb.numRegs, th, tl, tp_h, tp_l, tot);
  hipStreamDestroy(s1); hipStreamDestroy(s2);
}

int main() {
  const size_t n = 1ull << 30;   // 8 GiB read + 8 GiB written
  double *out, *src, *dst;
  CHECK(hipMalloc(&out, 256 * 4 * 512 * sizeof(double))); CHECK(hipMalloc(&src, n * 8)); CHECK(hipMalloc(&dst, n * 8));
  CHECK(hipMemset(src, 0, n * 8));
  const int iters = 6000;
  // heavy: ONE round of 256 resident workgroups for the whole run; light is queued right behind it on the other stream
  run(heavy216, stream80p, out, src, dst, n, iters * 4);   // (grid of the light kernel: see LGRID)
  run(heavy216, stream80, out, src, dst, n, iters * 4);
  run(heavy216, stream64p, out, src, dst, n, iters * 4);
  run(heavy216, light64, out, src, dst, n, iters);
  run(heavy216, light64p, out, src, dst, n, iters);
  run(heavy216, light80, out, src, dst, n, iters);
  run(heavy216, light80p, out, src, dst, n, iters);
  run(heavy216, light48p, out, src, dst, n, iters);
  run(heavy216, light32p, out, src, dst, n, iters);
  run(heavy216, light96, out, src, dst, n, iters);
  run(heavy128, light128, out, src, dst, n, iters);
  return 0;
}
