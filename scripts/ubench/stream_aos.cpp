// Would the 3D MHD update stream faster if the sweep wrote the three 5-component face fluxes cell-major (AoS: 40 contiguous bytes per
// cell and direction) instead of as 15 SoA components?  Same bytes: 8 U + 3 emf SoA read streams + 3 AoS-5 streams, 8 SoA write streams.
// Also: the plain few-stream copy rate of this GPU (2 R + 2 W) for reference.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int R, int W>
__global__ void __launch_bounds__(256) soa(const double* __restrict__ in, double* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double s = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) s += in[i + r * n];
#pragma unroll
  for (int w = 0; w < W; ++w) __builtin_nontemporal_store(s + w, &out[i + w * n]);
}
// 11 SoA streams + 3 AoS streams of 5 doubles per cell
__global__ void __launch_bounds__(256) aos(const double* __restrict__ in, double* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double s = 0;
#pragma unroll
  for (int r = 0; r < 11; ++r) s += in[i + r * n];
  const double* a = in + 11 * n;
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int v = 0; v < 5; ++v) s += a[(size_t)d * 5 * n + i * 5 + v];
#pragma unroll
  for (int w = 0; w < 8; ++w) __builtin_nontemporal_store(s + w, &out[i + w * n]);
}
template <class F>
float timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int it = 0; it < 5; ++it) f();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}
int main() {
  const size_t n = 518ull * 518 * 518;
  double *in, *out;
  if (hipMalloc(&in, n * 8 * 26) != hipSuccess || hipMalloc(&out, n * 8 * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemset(in, 0, n * 8 * 26);
  const unsigned g1 = (unsigned)((n + 255) / 256);
  const double gb = (double)n * 8 * 34 / 1e9;
  float a = timeit([&] { soa<26, 8><<<g1, 256>>>(in, out, n); });
  printf("26 R + 8 W SoA streams                                  : %.3f ms  %.2f TB/s\n", a, gb / a);
  a = timeit([&] { aos<<<g1, 256>>>(in, out, n); });
  printf("11 SoA + 3 x AoS-5 read streams, 8 W                    : %.3f ms  %.2f TB/s\n", a, gb / a);
  a = timeit([&] { soa<2, 2><<<g1, 256>>>(in, out, n); });
  printf(" 2 R + 2 W streams                                      : %.3f ms  %.2f TB/s\n", a, (double)n * 8 * 4 / 1e9 / a);
  a = timeit([&] { soa<8, 8><<<g1, 256>>>(in, out, n); });
  printf(" 8 R + 8 W streams                                      : %.3f ms  %.2f TB/s\n", a, (double)n * 8 * 16 / 1e9 / a);
  a = timeit([&] { soa<14, 8><<<g1, 256>>>(in, out, n); });
  printf("14 R + 8 W streams                                      : %.3f ms  %.2f TB/s\n", a, (double)n * 8 * 22 / 1e9 / a);
  return 0;
}
