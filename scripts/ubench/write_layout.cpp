// Does the layout of a 38-component write stream matter?  (a) SoA: 38 arrays of n doubles, a wave writes 38 separate
// 512-byte pieces; (b) blocked: per 64-cell block the 38 components are contiguous (a wave writes one 19.5 kB run);
// (c) one plain 8-byte stream of the same total size.  n = 518^3.
#include <hip/hip_runtime.h>
#include <cstdio>
const int W = 38;
__global__ void __launch_bounds__(256) soa(double* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
#pragma unroll
  for (int w = 0; w < W; ++w) __builtin_nontemporal_store((double)(i + w), &out[i + w * n]);
}
__global__ void __launch_bounds__(256) blocked(double* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double* o = out + (i >> 6) * (size_t)(W * 64) + (i & 63);
#pragma unroll
  for (int w = 0; w < W; ++w) __builtin_nontemporal_store((double)(i + w), &o[w * 64]);
}
__global__ void __launch_bounds__(256) plain(double* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) __builtin_nontemporal_store((double)i, &out[i]);
}
template <class F> float timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); for (int it = 0; it < 3; ++it) f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 3;
}
int main() {
  const size_t n = 518ull * 518 * 518, nb = (n + 63) / 64 * 64;
  double* out;
  if (hipMalloc(&out, nb * 8 * W) != hipSuccess) { printf("alloc failed\n"); return 1; }
  const unsigned grid = (unsigned)((n + 255) / 256);
  float a = timeit([&] { soa<<<grid, 256>>>(out, n); });
  float b = timeit([&] { blocked<<<grid, 256>>>(out, n); });
  const size_t big = n * W; const unsigned gridp = (unsigned)((big + 255) / 256);
  float c = timeit([&] { plain<<<gridp, 256>>>(out, big); });
  const double gb = n * 8.0 * W * 1e-9;
  printf("38-component write of %.1f GB: SoA %.2f ms (%.2f TB/s)  blocked %.2f ms (%.2f TB/s)  plain stream %.2f ms (%.2f TB/s)\n", gb, a, gb / a, b, gb / b, c, gb / c);
  return 0;
}
