// rg_backend.h (HIP / gfx950) -- the only backend of the product library.
// Thin layer between the step driver (rgpu_api.cpp) and the HIP runtime: kernel launch of per-cell functors on a
// flat 1D grid, a wave64 max-reduction for the CFL scan, device memory and event helpers.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstddef>
#include <cstdio>
#include <string>

#define RG_DEVFN __device__ __forceinline__
// promise to the optimiser about a launch-uniform parameter (kernels specialised at launch, see launchers.h)
#define RG_ASSUME(cond) __builtin_assume(cond)
#define RG_BACKEND_NAME "hip-gfx950"
// kernel launches return before the kernel has run: what a kernel leaves in device memory is not visible to the host code that queues
// the next launch (the test-only host emulation sets 1: its "launches" are host loops)
#define RG_SYNC_LAUNCH 0
// store of a value that is not read again before it has left the caches (T, F, emf, the new state): nontemporal,
// so that it does not evict the stencil neighbourhood the same XCD re-reads from its L2
#ifdef RG_NO_STREAM_STORE
#define RG_STREAM_STORE(ptr, val) (*(ptr) = (val))
#else
#define RG_STREAM_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#endif

namespace rgpu {

typedef hipStream_t rg_stream_t;
typedef hipEvent_t rg_event_t;

inline const char* rg_err_str(hipError_t e) { return hipGetErrorString(e); }

// ---- IEEE division / square root with the denominator work shared ----------------------------------------------
// hipcc expands an fp64 "n / d" into v_div_scale x2, v_rcp, two Newton steps on the reciprocal (4 FMA), the
// quotient (mul, FMA, v_div_fmas) and v_div_fixup.  Outside the exponent ranges where v_div_scale rescales
// (|d| or |n| beyond ~2^+-900, denormal quotients) the scale / fixup instructions are the identity and the result
// is the correctly rounded quotient produced by exactly the sequence below.  The Riemann solvers divide several
// numerators by the same denominator (66 divisions by 24 distinct denominators per edge EMF): rg_recip does the
// reciprocal refinement once, rg_div the 3-instruction quotient per numerator -- same bits as "/" for every value a
// simulation state can take, at 8 + 3(k-1) instead of 11k VALU instructions.  (Difference to "/": a -0 numerator
// over a positive denominator gives +0 instead of -0, and division by exactly 0 gives NaN instead of +-inf.)
//
// RG_ARITH_FAST (the librgpu_fast.so build, together with -ffp-contract=fast; see rgpu_arithmetic() in rgpu.h): the
// "contracted" arithmetic drops the last correction of each sequence -- one Newton step on the reciprocal, the plain product
// n * (1/d) as the quotient, one correction of the square root -- which leaves results within ~1 ulp instead of correctly
// rounded.  Measured on the golden fixtures: relative L2 to the reference <= 2e-14 (tolerance 1e-12).
struct rg_recip_t { double d, r; };
RG_DEVFN rg_recip_t rg_recip(double d) {
  rg_recip_t R;
  R.d = d;
  const double r0 = __builtin_amdgcn_rcp(d);
  const double e0 = __builtin_fma(-d, r0, 1.0);
  const double r1 = __builtin_fma(r0, e0, r0);
#ifdef RG_ARITH_FAST
  R.r = r1;
#else
  const double e1 = __builtin_fma(-d, r1, 1.0);
  R.r = __builtin_fma(r1, e1, r1);
#endif
  return R;
}
RG_DEVFN double rg_div(double n, const rg_recip_t& R) {
  const double q0 = n * R.r;
#ifdef RG_ARITH_FAST
  return q0;
#else
  const double rem = __builtin_fma(-R.d, q0, n);
  return __builtin_fma(rem, R.r, q0);
#endif
}
// Reciprocals of two / four denominators at once.  Exact arithmetic: each its own (rg_recip).  Contracted arithmetic: ONE v_rcp_f64 of
// the product and three / nine multiplications -- the transcendental unit runs at a quarter of the fp64 rate (16 cycles per wave
// against 4), so 1 / (a b) . b costs less than a second reciprocal; two to three more roundings per result (within the ~1-ulp class of
// this arithmetic).  For denominators that cannot vanish and whose product stays in range (densities, differences of signal speeds).
RG_DEVFN void rg_recip2(double a, double b, rg_recip_t& A, rg_recip_t& B) {
#ifdef RG_ARITH_FAST
  const double r = rg_recip(a * b).r;
  A.d = a; A.r = r * b;
  B.d = b; B.r = r * a;
#else
  A = rg_recip(a); B = rg_recip(b);
#endif
}
RG_DEVFN void rg_recip4(double a, double b, double c, double d, rg_recip_t& A, rg_recip_t& B, rg_recip_t& Cc, rg_recip_t& D) {
#ifdef RG_ARITH_FAST
  const double ab = a * b, cd = c * d;
  const double r = rg_recip(ab * cd).r;
  const double rab = r * cd, rcd = r * ab;
  A.d = a; A.r = rab * b;
  B.d = b; B.r = rab * a;
  Cc.d = c; Cc.r = rcd * d;
  D.d = d; D.r = rcd * c;
#else
  A = rg_recip(a); B = rg_recip(b); Cc = rg_recip(c); D = rg_recip(d);
#endif
}
// the compiler's sqrt minus its rescaling of arguments below 2^-767 (v_cmp, v_cndmask x2, v_ldexp x2)
RG_DEVFN double rg_sqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = y * 0.5;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double e = __builtin_fma(-g, g, x);
  g = __builtin_fma(e, h, g);
#ifndef RG_ARITH_FAST
  e = __builtin_fma(-g, g, x);
  g = __builtin_fma(e, h, g);
#endif
  return __builtin_amdgcn_class(x, 0x260) ? x : g;   // +-0 and +inf map to themselves
}

// same for an argument known to be positive and finite (densities, d2 + sqrt(..) of the fast speed): no +-0 / inf test
RG_DEVFN double rg_sqrt_pos(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = y * 0.5;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double e = __builtin_fma(-g, g, x);
  g = __builtin_fma(e, h, g);
#ifdef RG_ARITH_FAST
  return g;
#else
  e = __builtin_fma(-g, g, x);
  return __builtin_fma(e, h, g);
#endif
}

// root of d2^2 - c2 bn^2 / rho inside the fast magnetosonic speed: non-negative in real arithmetic, exactly 0 for a purely normal
// field at b^2 = rho c2.  Exact arithmetic: the reference's sqrt (0 -> 0, a negative round-off residue -> NaN like the reference).
// Contracted arithmetic: the radicand clamped to 2^-1000 from below (one v_max_f64 instead of a class test and two selects; the root
// of the clamp, 1e-151, is added to d2 > 0)
RG_DEVFN double rg_sqrt_radicand(double x) {
#ifdef RG_ARITH_FAST
  return rg_sqrt_pos(fmax(x, 0x1p-1000));
#else
  return rg_sqrt(x);
#endif
}

// reciprocal of sqrt(x), x positive and finite, for "n / sqrt(x)" (the Alfven speeds of the 2D HLLD solver divide by twelve
// such roots per edge).  Exact arithmetic: the correctly rounded root, then its shared reciprocal -- the reference's two
// operations.  Contracted arithmetic: one refined rsq.
RG_DEVFN rg_recip_t rg_recip_sqrt_pos(double x) {
#ifdef RG_ARITH_FAST
  rg_recip_t R;
  const double y = __builtin_amdgcn_rsq(x);
  const double t = x * y;                              // ~ sqrt(x)
  const double e = __builtin_fma(-t, y, 1.0);          // 1 - x y^2
  R.r = __builtin_fma(0.5 * y, e, y);
  R.d = t;
  return R;
#else
  return rg_recip(rg_sqrt_pos(x));
#endif
}

// true on every lane of the wave when the predicate holds on any active lane (wave-uniform branch conditions)
RG_DEVFN bool rg_wave_any(bool pred) { return __ballot(pred) != 0ull; }

// max of non-negative doubles into one of several device slots (the CFL scan that rides in the MHD update kernel): the
// plain read filters out almost every call once a slot holds a large value (a stale read can only cause a redundant atomic,
// never a missed one: stale values are <= the current one)
RG_DEVFN void rg_slot_max(unsigned long long* slot, double v) {
  if (v > __longlong_as_double((long long)*reinterpret_cast<volatile unsigned long long*>(slot))) atomicMax(slot, (unsigned long long)__double_as_longlong(v));
}
// the same with the maximum of the wave formed first (ds_bpermute butterfly; a lane that has left the kernel reads as 0, the
// neutral element here): one atomic per wave instead of up to 64 on one address.  Every lane still in the kernel must call it
// (v = 0 for cells that do not take part).
RG_DEVFN void rg_slot_max_wave(unsigned long long* slot, double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  const unsigned long long act = __ballot(1);   // after the xor butterfly every lane holds the maximum: the first active one stores it
  if (__lane_id() == (unsigned)(__ffsll((long long)act) - 1)) rg_slot_max(slot, v);
}
enum { RG_DT_SLOTS = 1024 };

// ---- flat per-cell kernels ---------------------------------------------------------------------------------
// One thread per array element, x fastest: every SoA component load / store of a wave is one contiguous
// 512-byte segment.  BLOCK is a multiple of the 64-lane wavefront.
// MINW = minimum waves per SIMD the register allocator must leave room for (__launch_bounds__ 2nd argument).
template <int BLOCK, class K, int MINW = 1>
__global__ void __launch_bounds__(BLOCK, MINW) rg_kernel(unsigned n, K k) {
  const unsigned idx = blockIdx.x * (unsigned)BLOCK + threadIdx.x;
  if (idx < n) k(idx);
}

template <int BLOCK, int MINW = 1, class K>
inline int rg_launch(rg_stream_t s, unsigned n, const K& k) {
  if (n == 0) return 0;
  const unsigned grid = (n + BLOCK - 1) / BLOCK;
  hipLaunchKernelGGL((rg_kernel<BLOCK, K, MINW>), dim3(grid), dim3(BLOCK), 0, s, n, k);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// same body over the flat index range [idx0, idx0 + n): used by the z-chunked, two-stream schedule
template <int BLOCK, class K, int MINW = 1>
__global__ void __launch_bounds__(BLOCK, MINW) rg_kernel_range(unsigned idx0, unsigned n, K k) {
  const unsigned off = blockIdx.x * (unsigned)BLOCK + threadIdx.x;
  if (off < n) k(idx0 + off);
}

template <int BLOCK, int MINW = 1, class K>
inline int rg_launch_range(rg_stream_t s, unsigned idx0, unsigned n, const K& k) {
  if (n == 0) return 0;
  const unsigned grid = (n + BLOCK - 1) / BLOCK;
  hipLaunchKernelGGL((rg_kernel_range<BLOCK, K, MINW>), dim3(grid), dim3(BLOCK), 0, s, idx0, n, k);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// same body over whole z planes [idx0 + p * plane_cells, +plane_cells), p < nplanes, with an XCD-aware order.
// The dispatcher deals workgroup b to XCD (b mod 8), each XCD with its own 4 MB L2.  With the linear mapping a
// cell's y and z stencil neighbours are fetched by other XCDs, so every stencil re-read misses L2 and goes to
// Infinity Cache / HBM.  Here XCD x owns the y band x of every plane (band = 1/8 of the plane's workgroups) and
// walks it in sub-bands of T workgroups: sub-band s of plane 0, 1, ... nplanes-1, then sub-band s+1.  The +-1 row
// and +-1 plane neighbours of a sub-band were touched by the same XCD a few hundred workgroups earlier: the
// working set (3 planes x sub-band x ~14 components ~ 1.5 MB at T*BLOCK = 4096 cells) stays in that XCD's L2.
struct rg_plane_map {
  unsigned idx0, plane_cells, nplanes, band, T;   // band = workgroups per XCD and plane, T = workgroups per sub-band
};
template <int BLOCK, class K, int MINW = 1>
__global__ void __launch_bounds__(BLOCK, MINW) rg_kernel_planes(rg_plane_map m, K k) {
  const unsigned b = blockIdx.x;
  const unsigned xcd = b & 7u, slot = b >> 3;
  const unsigned per_sub = m.nplanes * m.T;
  const unsigned s = slot / per_sub, r = slot - s * per_sub;
  const unsigned p = r / m.T, t = r - p * m.T;
  const unsigned in_band = s * m.T + t;
  const unsigned off = (xcd * m.band + in_band) * (unsigned)BLOCK + threadIdx.x;
  if (in_band < m.band && off < m.plane_cells) k(m.idx0 + p * m.plane_cells + off);
}
// sub = sub-band size in cells (per context, rgpu_ctx::xcd_sub); 0: linear order
template <int BLOCK, int MINW = 1, class K>
inline int rg_launch_planes(rg_stream_t s, unsigned idx0, unsigned plane_cells, unsigned nplanes, const K& k, unsigned sub) {
  if (nplanes == 0 || plane_cells == 0) return 0;
  if (sub == 0) return rg_launch_range<BLOCK, MINW>(s, idx0, plane_cells * nplanes, k);
  rg_plane_map m;
  m.idx0 = idx0; m.plane_cells = plane_cells; m.nplanes = nplanes;
  const unsigned bpp = (plane_cells + BLOCK - 1) / BLOCK;
  m.band = (bpp + 7) / 8;
  unsigned T = sub / BLOCK > 0 ? sub / BLOCK : 1;
  if (T > m.band) T = m.band;
  const unsigned nsub = (m.band + T - 1) / T;
  m.T = (m.band + nsub - 1) / nsub;   // equal sub-bands: at most nsub-1 idle workgroup slots per band
  const unsigned grid = 8u * nsub * nplanes * m.T;
  hipLaunchKernelGGL((rg_kernel_planes<BLOCK, K, MINW>), dim3(grid), dim3(BLOCK), 0, s, m, k);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- max reduction (CFL scan) ---------------------------------------------------------------------------------
// Grid-stride accumulation, wave64 butterfly with __shfl_down, one LDS slot per wave, one 64-bit atomicMax per
// block.  All values are >= 0, so the IEEE bit pattern orders like an unsigned integer; max is exact and
// order-independent, hence bit-identical to the reference's sequential scan.
template <int BLOCK, class K>
__global__ void __launch_bounds__(BLOCK) rg_reduce_max_kernel(unsigned idx0, unsigned n, K k, unsigned long long* out) {
  double v = 0.0;
  for (unsigned off = blockIdx.x * (unsigned)BLOCK + threadIdx.x; off < n; off += gridDim.x * (unsigned)BLOCK)
    v = fmax(v, k(idx0 + off));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  __shared__ double wave_max[BLOCK / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_max[wave] = v;
  __syncthreads();
  if (wave == 0) {
    v = (lane < BLOCK / 64) ? wave_max[lane] : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    if (lane == 0) atomicMax(out, (unsigned long long)__double_as_longlong(v));
  }
}

// max over the flat index range [idx0, idx0 + n); reset = false accumulates into the value already in *d_out
template <class K>
inline int rg_reduce_max(rg_stream_t s, unsigned n, const K& k, unsigned long long* d_out, unsigned idx0 = 0, bool reset = true) {
  const int BLOCK = 256;
  unsigned grid = (n + BLOCK - 1) / BLOCK;
  if (grid > 2048u) grid = 2048u;  // 256 CUs x 8 resident blocks; the rest is grid-strided
  if (grid == 0) grid = 1;
  if (reset && hipMemsetAsync(d_out, 0, sizeof(unsigned long long), s) != hipSuccess) return -1;
  if (n == 0) return 0;
  hipLaunchKernelGGL((rg_reduce_max_kernel<BLOCK, K>), dim3(grid), dim3(BLOCK), 0, s, idx0, n, k, d_out);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- memory / stream helpers ------------------------------------------------------------------------------------
inline int rg_device_count() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
inline int rg_current_device() { int d = 0; return hipGetDevice(&d) == hipSuccess ? d : -1; }
inline void rg_set_device(int d) { (void)hipSetDevice(d); }
inline int rg_pointer_device(const void* p) {   // device owning a device pointer, -1 if unknown
  hipPointerAttribute_t a;
  if (!p || hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return -1; }
  return a.type == hipMemoryTypeDevice ? a.device : -1;
}
inline int rg_malloc(void** p, size_t bytes) { return hipMalloc(p, bytes) == hipSuccess ? 0 : -1; }
inline void rg_free(void* p) { if (p) (void)hipFree(p); }
inline int rg_host_alloc(void** p, size_t bytes) { return hipHostMalloc(p, bytes, hipHostMallocDefault) == hipSuccess ? 0 : -1; }
inline void rg_host_free(void* p) { if (p) (void)hipHostFree(p); }
inline int rg_memset_async(void* p, int v, size_t bytes, rg_stream_t s) { return hipMemsetAsync(p, v, bytes, s) == hipSuccess ? 0 : -1; }
inline int rg_copy_h2d(void* d, const void* h, size_t bytes, rg_stream_t s) { return hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s) == hipSuccess ? 0 : -1; }
inline int rg_copy_d2h(void* h, const void* d, size_t bytes, rg_stream_t s) { return hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s) == hipSuccess ? 0 : -1; }
inline int rg_copy_d2d(void* d, const void* s_, size_t bytes, rg_stream_t s) { return hipMemcpyAsync(d, s_, bytes, hipMemcpyDeviceToDevice, s) == hipSuccess ? 0 : -1; }
inline rg_stream_t rg_stream_from_handle(void* h) { return (hipStream_t)h; }
inline void* rg_stream_to_handle(rg_stream_t s) { return (void*)s; }
inline int rg_stream_sync(rg_stream_t s) { return hipStreamSynchronize(s) == hipSuccess ? 0 : -1; }
inline const char* rg_last_error_string() { return hipGetErrorString(hipGetLastError()); }

// second stream + ordering-only events for overlapping HBM-bound and VALU-bound kernels of one step
// prio: -1 = lowest available priority, 0 = default, +1 = highest.  The VALU-bound Riemann kernels go to a LOW
// priority queue: their long-lived waves otherwise take over the wave slots and starve the short HBM-bound kernels
// that run next to them (measured: prim/elec 3-7x slower when co-scheduled at equal priority).
inline int rg_stream_create(rg_stream_t* s, int prio = 0) {
  int least = 0, greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { least = greatest = 0; }
  const int p = prio < 0 ? least : prio > 0 ? greatest : 0;
  return hipStreamCreateWithPriority(s, hipStreamNonBlocking, p) == hipSuccess ? 0 : -1;
}
inline void rg_stream_destroy(rg_stream_t s) { if (s) (void)hipStreamDestroy(s); }
inline int rg_order_event_create(rg_event_t* e) { return hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess ? 0 : -1; }
inline int rg_stream_wait_event(rg_stream_t s, rg_event_t e) { return hipStreamWaitEvent(s, e, 0) == hipSuccess ? 0 : -1; }

inline int rg_event_create(rg_event_t* e) { return hipEventCreate(e) == hipSuccess ? 0 : -1; }
inline void rg_event_destroy(rg_event_t e) { (void)hipEventDestroy(e); }
inline int rg_event_record(rg_event_t e, rg_stream_t s) { return hipEventRecord(e, s) == hipSuccess ? 0 : -1; }
inline double rg_event_elapsed_ms(rg_event_t a, rg_event_t b) {
  float ms = 0.f;
  if (hipEventSynchronize(b) != hipSuccess) return 0.0;
  if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return 0.0;
  return (double)ms;
}

}  // namespace rgpu
