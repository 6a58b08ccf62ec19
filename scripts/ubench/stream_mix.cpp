// HBM bandwidth of SoA streaming with R read streams and W write streams per element (the shape of the step's
// "memory" kernels: trace reads ~14 component arrays and writes 38).  n = 518^3 elements, component stride n.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int R, int W>
__global__ void __launch_bounds__(256) k(const double* __restrict__ in, double* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double s = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) s += in[i + r * n];
#pragma unroll
  for (int w = 0; w < W; ++w) out[i + w * n] = s + w;
}
template <int R, int W>
void run(const double* in, double* out, size_t n) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const unsigned grid = (unsigned)((n + 255) / 256);
  k<R, W><<<grid, 256>>>(in, out, n);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int it = 0; it < 3; ++it) k<R, W><<<grid, 256>>>(in, out, n);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  printf("R=%2d W=%2d  %7.3f ms  %6.2f TB/s  (read %5.2f, write %5.2f TB/s)\n", R, W, ms, (R + W) * 8.0 * n / ms * 1e-9,
         R * 8.0 * n / ms * 1e-9, W * 8.0 * n / ms * 1e-9);
}
int main() {
  const size_t n = 518ull * 518 * 518;
  double *in, *out;
  if (hipMalloc(&in, n * 8 * 26) != hipSuccess || hipMalloc(&out, n * 8 * 38) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemset(in, 0, n * 8 * 26);
  run<1, 1>(in, out, n); run<8, 8>(in, out, n); run<16, 0 + 1>(in, out, n); run<26, 8>(in, out, n);
  run<1, 16>(in, out, n); run<1, 38>(in, out, n); run<14, 38>(in, out, n); run<8, 3>(in, out, n);
  return 0;
}
