// step_clock.h (HIP / gfx950 only) -- the time step on the device: lets a run of 2D steps be queued without a host round trip.
//
// A 2D step of the fused kernels (hip/tiled_mhd2d.h, hip/tiled_hydro2d.h) is ONE launch of 20-50 us that leaves the CFL maxima of
// the state it wrote in RG_DT_SLOTS device slots and, in a box of periodic / reflecting / outflow faces, that state's ghost cells.
// What the reference's loop does between two steps (HydroRunBase::start, MHDRunGodunov.cpp:3921-3990: dt = cfl / max(1/dt), t += dt,
// "while t < tEnd") then costs more than the step: an 8 KB read-back + host synchronisation, a memset launch, ~15-20 us of idle GPU.
// step_clock_kernel does it on the device: one workgroup folds the slots (and re-zeroes them), forms dt, dt/dx, dt/dy with the
// host's expressions (IEEE division: the same doubles), advances t, evaluates the loop condition, and leaves a StepClock record
// that the step kernel reads instead of its by-value dt arguments.  The host reads the records of a whole batch afterwards.
#pragma once
#include "tiled_hydro.h"

namespace rgpu_tiled {

// one record per step of a batch.  stop: 0 = the step runs; 1 = t >= tEnd before this step (it and all later steps of the batch are
// no-ops); 2 = dt is not a number (same).
struct StepClock { double dt, dtdx, dtdy, t_next; int stop, pad; };

__global__ void __launch_bounds__(1024) step_clock_kernel(unsigned long long* __restrict__ slots, double cfl, double seed, double dx, double dy,
                                                        double t0, double tEnd, const StepClock* prev, StepClock* out) {
  __shared__ double red[16];
  const int t = (int)threadIdx.x;
  static_assert(rgpu::RG_DT_SLOTS == 1024, "one slot per thread");
  double v = __longlong_as_double((long long)slots[t]);
  slots[t] = 0ull;   // the step kernel that follows accumulates the maxima of the state it writes
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  if ((t & 63) == 0) red[t >> 6] = v;
  __syncthreads();
  if (t == 0) {
    double m = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) m = fmax(m, red[w]);
    const double tcur = prev ? prev->t_next : t0;
    int stop = prev ? prev->stop : 0;
    if (!stop && !(tcur < tEnd)) stop = 1;
    // rgpu_compute_inv_dt + rgpu_compute_dt on the host: v = max(slots), MHD: v = max(v, smallc / min(dx, dy)); dt = cfl / v
    const double inv = fmax(m, seed);
    const double dt = cfl / inv;
    if (!stop && !(dt == dt)) stop = 2;
    out->stop = stop; out->pad = 0;
    out->dt = stop ? 0.0 : dt;
    out->dtdx = stop ? 0.0 : dt / dx;
    out->dtdy = stop ? 0.0 : dt / dy;
    out->t_next = stop ? tcur : tcur + dt;
  }
}

inline bool step_clock_supported() { return tiled_enabled() && !std::getenv("RGPU_NO_STEP_CLOCK"); }
inline int launch_step_clock(rg_stream_t s, unsigned long long* slots, double cfl, double seed, double dx, double dy, double t0, double tEnd,
                             const StepClock* prev, StepClock* out) {
  hipLaunchKernelGGL(step_clock_kernel, dim3(1), dim3(1024), 0, s, slots, cfl, seed, dx, dy, t0, tEnd, prev, out);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace rgpu_tiled
