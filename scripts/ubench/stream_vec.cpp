// Does packing two components per 16-byte element help the many-stream pattern of the trace kernel?
// R read streams / W write streams of double (SoA) against R/2, W/2 streams of double2.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int R, int W>
__global__ void __launch_bounds__(256) k1(const double* __restrict__ in, double* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double s = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) s += in[i + r * n];
#pragma unroll
  for (int w = 0; w < W; ++w) __builtin_nontemporal_store(s + w, &out[i + w * n]);
}
template <int R, int W>
__global__ void __launch_bounds__(256) k2(const double2* __restrict__ in, double2* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double s = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) { const double2 v = in[i + r * n]; s += v.x + v.y; }
#pragma unroll
  for (int w = 0; w < W; ++w) { double2 v; v.x = s + w; v.y = s - w; out[i + w * n] = v; }
}
template <class F>
float timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int it = 0; it < 3; ++it) f();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 3;
}
int main() {
  const size_t n = 518ull * 518 * 518;
  double *in, *out;
  if (hipMalloc(&in, n * 8 * 16) != hipSuccess || hipMalloc(&out, n * 8 * 38) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemset(in, 0, n * 8 * 16);
  const unsigned grid = (unsigned)((n + 255) / 256);
  float a = timeit([&] { k1<14, 38><<<grid, 256>>>(in, out, n); });
  float b = timeit([&] { k2<7, 19><<<grid, 256>>>((const double2*)in, (double2*)out, n); });
  printf("14 r + 38 w double  : %7.3f ms %5.2f TB/s\n", a, 52 * 8.0 * n / a * 1e-9);
  printf(" 7 r + 19 w double2 : %7.3f ms %5.2f TB/s\n", b, 52 * 8.0 * n / b * 1e-9);
  a = timeit([&] { k1<16, 8><<<grid, 256>>>(in, out, n); });
  b = timeit([&] { k2<8, 4><<<grid, 256>>>((const double2*)in, (double2*)out, n); });
  printf("16 r +  8 w double  : %7.3f ms %5.2f TB/s\n", a, 24 * 8.0 * n / a * 1e-9);
  printf(" 8 r +  4 w double2 : %7.3f ms %5.2f TB/s\n", b, 24 * 8.0 * n / b * 1e-9);
  return 0;
}
