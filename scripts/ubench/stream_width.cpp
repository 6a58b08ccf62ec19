// The 3D MHD update kernel streams 26 read + 8 write SoA components, one cell (8 B per stream) per lane.  Does a lane that owns
// TWO adjacent cells (16-byte loads / stores) reach a higher bandwidth on the same byte count?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int R, int W>
__global__ void __launch_bounds__(256) k8(const double* __restrict__ in, double* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double s = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) s += in[i + r * n];
#pragma unroll
  for (int w = 0; w < W; ++w) __builtin_nontemporal_store(s + w, &out[i + w * n]);
}
template <int R, int W>
__global__ void __launch_bounds__(256) k16(const double* __restrict__ in, double* __restrict__ out, size_t n) {
  const size_t i = 2 * ((size_t)blockIdx.x * 256 + threadIdx.x);
  if (i + 1 >= n) return;
  double s0 = 0, s1 = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) { const double2 v = *reinterpret_cast<const double2*>(in + i + r * n); s0 += v.x; s1 += v.y; }
#pragma unroll
  for (int w = 0; w < W; ++w) { double2 v; v.x = s0 + w; v.y = s1 - w; __builtin_nontemporal_store(v.x, &out[i + w * n]); __builtin_nontemporal_store(v.y, &out[i + 1 + w * n]); }
}
template <int R, int W>
__global__ void __launch_bounds__(256) k16v(const double* __restrict__ in, double* __restrict__ out, size_t n) {
  const size_t i = 2 * ((size_t)blockIdx.x * 256 + threadIdx.x);
  if (i + 1 >= n) return;
  double s0 = 0, s1 = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) { const double2 v = *reinterpret_cast<const double2*>(in + i + r * n); s0 += v.x; s1 += v.y; }
#pragma unroll
  for (int w = 0; w < W; ++w) { double2 v; v.x = s0 + w; v.y = s1 - w; *reinterpret_cast<double2*>(out + i + w * n) = v; }
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
  const size_t n = 518ull * 518 * 518;      // even
  double *in, *out;
  if (hipMalloc(&in, n * 8 * 26) != hipSuccess || hipMalloc(&out, n * 8 * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemset(in, 0, n * 8 * 26);
  const unsigned g1 = (unsigned)((n + 255) / 256), g2 = (unsigned)((n / 2 + 255) / 256);
  const double gb = (double)n * 8 * 34 / 1e9;
  float a = timeit([&] { k8<26, 8><<<g1, 256>>>(in, out, n); });
  printf("26 R + 8 W streams,  8 B per lane                     : %.3f ms  %.2f TB/s\n", a, gb / a);
  a = timeit([&] { k16<26, 8><<<g2, 256>>>(in, out, n); });
  printf("26 R + 8 W streams, 16 B loads, 2 x 8 B nt stores     : %.3f ms  %.2f TB/s\n", a, gb / a);
  a = timeit([&] { k16v<26, 8><<<g2, 256>>>(in, out, n); });
  printf("26 R + 8 W streams, 16 B loads, 16 B stores           : %.3f ms  %.2f TB/s\n", a, gb / a);
  return 0;
}
