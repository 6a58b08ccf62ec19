// rg_tiled.h (TEST-ONLY host emulation) -- NOT part of the product.  The LDS-tiled cooperative kernels exist in the HIP
// backend only (workgroup barriers and LDS have no host-loop equivalent): the emulation build always answers "not
// covered", so the step driver runs the flat per-cell kernels, which the CPU tests check as a second implementation.
#pragma once
namespace rgpu_tiled {
struct StepClock { double dt, dtdx, dtdy, t_next; int stop, pad; };   // (hip/step_clock.h; never used here)
inline bool step_clock_supported() { return false; }
inline int launch_step_clock(rgpu::rg_stream_t, unsigned long long*, double, double, double, double, double, double, const StepClock*, StepClock*) { return -1; }
inline int hydro3d_sweep(rgpu::rg_stream_t, const rgpu_dev::DevParams&, const double*, double*, double, double, double, int, int, unsigned long long* = 0) { return 1; }
inline bool mhd3d_sweep_covers(const rgpu_dev::DevParams&) { return false; }
inline bool hydro3d_sweep_covers(const rgpu_dev::DevParams&) { return false; }
inline bool mhd2d_step_covers(const rgpu_dev::DevParams&) { return false; }
inline int hydro2d_step(rgpu::rg_stream_t, const rgpu_dev::DevParams&, const double*, double*, double, double, unsigned long long*, int, const StepClock* = 0) { return 1; }
template <int SPEC_MRI, int SPEC_PLAIN>
inline int mhd3d_sweep(rgpu::rg_stream_t, const rgpu_dev::DevParams&, int, const double*, double*, double*,
                       double, double, double, double, int, int, int = 0) { return 1; }
template <int SPEC_PLAIN>
inline int mhd2d_step(rgpu::rg_stream_t, const rgpu_dev::DevParams&, const rgpu_dev::RotCoef&, bool, const double*, double*, double,
                      unsigned long long*, int, const StepClock* = 0) { return 1; }
}  // namespace rgpu_tiled
