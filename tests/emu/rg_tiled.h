// rg_tiled.h (TEST-ONLY host emulation) -- NOT part of the product.  The LDS-tiled cooperative kernels exist in the HIP
// backend only (workgroup barriers and LDS have no host-loop equivalent): the emulation build always answers "not
// covered", so the step driver runs the flat per-cell kernels, which the CPU tests check as a second implementation.
#pragma once
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "step_clock_rec.h"
namespace rgpu_tiled {
// the device-side time step (hip/step_clock.h): the same record (csrc/step_clock_rec.h), the fold of the slots as a host loop -- so
// that the batch logic of rgpu_run_steps / rgpu_comm_run_steps runs in the CPU tests (option step_clock = 0: off, as in the product)
using rgpu_dev::StepClock;
using rgpu_dev::ClockConst;
inline bool step_clock_supported() { return rgpu::options().step_clock != 0; }
inline int launch_step_clock(rgpu::rg_stream_t, unsigned long long* slots, const ClockConst& k, double t0, double tEnd, const StepClock* prev, StepClock* out) {
  double m = 0.0;
  for (int s = 0; s < (int)rgpu::RG_DT_SLOTS; ++s) { double v; std::memcpy(&v, slots + s, sizeof(v)); m = std::fmax(m, v); }
  rgpu_dev::step_clock_form(k, m, prev ? prev->t_next : t0, tEnd, prev ? prev->stop : 0, out);
  if (out->stop == 0) for (int s = 0; s < (int)rgpu::RG_DT_SLOTS; ++s) slots[s] = 0ull;   // (a stopped step keeps the maxima of the last state written)
  return 0;
}
inline int hydro3d_sweep(rgpu::rg_stream_t, const rgpu_dev::DevParams&, const double*, double*, double, double, double, int, int, unsigned long long* = 0, const StepClock* = 0, int = 0, int = 0) { return 1; }
inline bool mhd3d_sweep_covers(const rgpu_dev::DevParams&) { return false; }
inline bool hydro3d_sweep_covers(const rgpu_dev::DevParams&) { return false; }
inline bool mhd2d_step_covers(const rgpu_dev::DevParams&) { return false; }
inline int hydro2d_step(rgpu::rg_stream_t, const rgpu_dev::DevParams&, const double*, double*, double, double, unsigned long long*, int, const StepClock* = 0, const rgpu_dev::ClockFold* = 0) { return 1; }
inline bool step_clock_fold_enabled() { return false; }
template <int SPEC_MRI, int SPEC_PLAIN>
inline int mhd3d_sweep(rgpu::rg_stream_t, const rgpu_dev::DevParams&, int, const double*, double*, double*,
                       double, double, double, double, int, int, int = 0, const StepClock* = 0, double* = 0, int = 0) { return 1; }
template <int SPEC_PLAIN>
inline int mhd2d_step(rgpu::rg_stream_t, const rgpu_dev::DevParams&, const rgpu_dev::RotCoef&, bool, const double*, double*, double,
                      unsigned long long*, int, const StepClock* = 0) { return 1; }
}  // namespace rgpu_tiled
