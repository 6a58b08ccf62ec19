// step_clock.h (HIP / gfx950 only) -- the time step on the device: lets a run of steps be queued without a host round trip.
//
// A 2D step of the fused kernels (hip/tiled_mhd2d.h, hip/tiled_hydro2d.h) is ONE launch of 20-50 us that leaves the CFL maxima of
// the state it wrote in RG_DT_SLOTS device slots and, in a box of periodic / reflecting / outflow faces, that state's ghost cells.
// What the reference's loop does between two steps (HydroRunBase::start, MHDRunGodunov.cpp:3921-3990: dt = cfl / max(1/dt), t += dt,
// "while t < tEnd") then costs more than the step: an 8 KB read-back + host synchronisation, a memset launch, ~15-20 us of idle GPU.
// step_clock_kernel does it on the device: one workgroup folds the slots (and re-zeroes them), forms dt, dt/dx, dt/dy with the
// host's expressions (IEEE division: the same doubles), advances t, evaluates the loop condition, and leaves a StepClock record
// that the step kernel reads instead of its by-value dt arguments.  The host reads the records of a whole batch afterwards.
// Round 5: the record also carries what the 3D steps take from the host (dt/dz, rotating-frame coefficients, shearing-box offsets),
// so that the 3D hydro and MHD sweeps, the update and the shearing ghost fill read it too (csrc/step_clock_rec.h), and the z-slab
// driver queues batches of steps: all-reduce of the slots in place -> this kernel -> the step pieces, no host turn in between.
#pragma once
#include "tiled_hydro.h"
#include "../step_clock_rec.h"

namespace rgpu_tiled {

// One workgroup: folds the RG_DT_SLOTS CFL maxima (and re-zeroes them: the step kernels that follow accumulate the maxima of the
// state they write), then thread 0 forms the record of the step (step_clock_rec.h: the host's expressions in the host's order).
// prev: the previous record of the batch (0: the batch starts at t0).
__global__ void __launch_bounds__(1024) step_clock_kernel(unsigned long long* __restrict__ slots, ClockConst k, double t0, double tEnd,
                                                        const StepClock* prev, StepClock* out) {
  __shared__ double red[16];
  __shared__ int runs;
  const int t = (int)threadIdx.x;
  static_assert(rgpu::RG_DT_SLOTS == 1024, "one slot per thread");
  double v = __longlong_as_double((long long)slots[t]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  if ((t & 63) == 0) red[t >> 6] = v;
  __syncthreads();
  if (t == 0) {
    double m = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) m = fmax(m, red[w]);
    StepClock r;
    step_clock_form(k, m, prev ? prev->t_next : t0, tEnd, prev ? prev->stop : 0, &r);
    *out = r;
    runs = r.stop == 0;
  }
  __syncthreads();
  // a step that runs accumulates the maxima of the state it writes into zeroed slots; a stopped one (and every step behind it) is a
  // no-op and leaves the slots as they are: after the batch they still hold the maxima of the last state written
  if (runs) slots[t] = 0ull;
}

// the fold of csrc/step_clock_rec.h (ClockFold) for a workgroup of NT threads; red: NT / 64 doubles of LDS of its own.  Contains one
// workgroup barrier.  Returns the record of this step (the same on every thread of every workgroup).
template <int NT>
__device__ __forceinline__ StepClock clock_fold(const ClockFold& f, double* red) {
  static_assert(NT % 64 == 0 && rgpu::RG_DT_SLOTS % NT == 0, "whole waves, whole trips over the slots");
  const int t = (int)threadIdx.x;
  double v = 0.0;
#pragma unroll
  for (int s = 0; s < rgpu::RG_DT_SLOTS / NT; ++s) v = fmax(v, __longlong_as_double((long long)f.in[s * NT + t]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  if ((t & 63) == 0) red[t >> 6] = v;
  __syncthreads();
  double m = red[0];
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) m = fmax(m, red[w]);
  StepClock r;
  step_clock_form(f.k, m, f.prev ? f.prev->t_next : f.t0, f.tEnd, f.prev ? f.prev->stop : 0, &r);
  if (blockIdx.x == 0) {
    if (t == 0) *f.out = r;
    if (!r.stop) {   // (a stopped step leaves every slot array as it is)
#pragma unroll
      for (int s = 0; s < rgpu::RG_DT_SLOTS / NT; ++s) f.zero[s * NT + t] = 0ull;
    }
  }
  return r;
}

// a double that every lane of the wave holds alike, moved to scalar registers (the record formed by clock_fold is computed on the
// vector unit; the kernels keep dt, dt/dx .. for their whole life, and their vector register file is what limits them)
__device__ __forceinline__ double rg_uniform(double x) {
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll)), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}

inline bool step_clock_fold_enabled() { return true; }
inline bool step_clock_supported() { return tiled_enabled() && rgpu::options().step_clock != 0; }
inline int launch_step_clock(rg_stream_t s, unsigned long long* slots, const ClockConst& k, double t0, double tEnd, const StepClock* prev, StepClock* out) {
  hipLaunchKernelGGL(step_clock_kernel, dim3(1), dim3(1024), 0, s, slots, k, t0, tEnd, prev, out);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace rgpu_tiled
