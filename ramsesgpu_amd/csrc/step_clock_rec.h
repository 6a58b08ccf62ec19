// step_clock_rec.h -- the time step as a DEVICE record: what the reference's loop forms on the host between two steps
// (MHDRunGodunov.cpp:3921-3990 `while (t < tEnd) { dt = compute_dt(); godunov_unsplit(nStep, dt); t += dt; }`, compute_dt_mhd
// MHDRunBase.cpp:140-250) evaluated by one small kernel from the CFL maxima the step kernels left in the context's device slots,
// so that a batch of steps can be queued without a host round trip (rgpu_run_steps, rgpu_comm_run_steps).
//
// Everything a step kernel takes from the host as a function of dt and t is in the record, formed with the host's expressions in
// the host's order (IEEE division, exact fmod; contraction off in both builds of the library), hence the same doubles:
//   dt, dt/dx, dt/dy, dt/dz                                               rgpu_compute_dt, hydro_core / mhd3d_core
//   rotating-frame coefficients lambda, ratio, alpha1, alpha2             rot_coef (MHDRunGodunov.cpp:2039-2053)
//   shearing-box flux / emf remap offsets at t + dt/2                     mhd3d_core (MHDRunGodunov.cpp:3213-3216)
//   shearing-box ghost remap offsets at t + dt                            do_make_boundaries_shear (MHDRunGodunov.cpp:3554-3557)
// Shared by the HIP backend (hip/step_clock.h: the kernel) and the test-only host emulation (tests/emu/rg_tiled.h: a host loop),
// and read by the step kernels through `const StepClock* clk` (0: the kernel takes its by-value arguments).
#pragma once

namespace rgpu_dev {

// stop: 0 = the step runs; 1 = t >= tEnd before this step; 2 = dt is not a number; 3 = 1/dt is not finite (a slab rank reported a
// failure through the all-reduce, or the solution blew up).  A stopped record turns its step and all later steps of the batch
// into no-ops: the reference's "first step that carries t past tEnd is the last" holds exactly.
struct StepClock {
  double dt, dtdx, dtdy, dtdz;
  double t_cur, t_next;
  double lambda, ratio, alpha1, alpha2;                 // RotCoef
  double remap_eps_min, remap_eps_max;                  // ShearRemap (t + dt/2)
  double ghost_eps_min, ghost_eps_max;                  // ShearGhost (t + dt)
  int remap_jplus, ghost_jplus;
  int stop, pad;
};

// what the record is formed from besides the folded maximum: constants of the run
struct ClockConst {
  double cfl, seed;             // seed: the floor of 1/dt (MHD: smallc / min(dx, dy); jet: (ujet + cjet) / dx), inv_dt_fetch
  double dx, dy, dz;
  double Omega0, xlen, ylen;    // rotating frame: Omega0 > 0; shearing box: xlen = dx * nx, ylen = dy * ny
  int rot, shear;
};

// The clock of a step INSIDE the step kernel (the fused 2D steps, 20-50 us each, where a clock kernel of its own is 10 % of the step):
// every workgroup folds the CFL maxima of the input state itself -- `in`, RG_DT_SLOTS values, L2-resident -- and forms the same
// record (the maximum is exact, so all workgroups get the same doubles); workgroup 0 writes it to `out` (for the host, and for the
// next step's t) and zeroes `zero`, the slot array the NEXT step accumulates into.  Three slot arrays rotate: step n reads S[n % 3],
// accumulates the maxima of the state it writes into S[(n + 1) % 3] (zeroed by step n - 1) and zeroes S[(n + 2) % 3] (read by step
// n - 1, which is complete).  out == 0: no fold (the kernel takes its by-value arguments, or a record through `clk`).
struct ClockFold {
  const StepClock* prev;            // record of the previous step of the batch (0: the batch starts at t0)
  StepClock* out;
  const unsigned long long* in;
  unsigned long long* zero;
  ClockConst k;
  double t0, tEnd;
};

#if defined(__clang__)
#pragma clang fp contract(off)
#endif
// max_inv: the folded CFL maximum; t_cur / prev_stop: from the previous record of the batch (or the batch's start time, 0)
RG_DEVFN void step_clock_form(const ClockConst& k, double max_inv, double t_cur, double tEnd, int prev_stop, StepClock* out) {
  int stop = prev_stop;
  if (!stop && !(t_cur < tEnd)) stop = 1;
  const double inv = fmax(max_inv, k.seed);
  const double dt = k.cfl / inv;
  if (!stop && !(dt == dt)) stop = 2;
  if (!stop && !(inv < __builtin_huge_val())) stop = 3;
  StepClock r;
  r.stop = stop; r.pad = 0;
  r.t_cur = t_cur;
  r.dt = stop ? 0.0 : dt;
  r.dtdx = stop ? 0.0 : dt / k.dx;
  r.dtdy = stop ? 0.0 : dt / k.dy;
  r.dtdz = stop ? 0.0 : dt / k.dz;
  r.t_next = stop ? t_cur : t_cur + dt;
  r.lambda = 0.0; r.ratio = 1.0; r.alpha1 = 1.0; r.alpha2 = 0.0;
  r.remap_jplus = 0; r.remap_eps_min = 0.0; r.remap_eps_max = 0.0;
  r.ghost_jplus = 0; r.ghost_eps_min = 0.0; r.ghost_eps_max = 0.0;
  if (!stop && k.rot) {
    double lambda = k.Omega0 * dt;
    lambda = 0.25 * lambda * lambda;
    r.lambda = lambda;
    r.ratio = (1.0 - lambda) / (1.0 + lambda);
    r.alpha1 = 1.0 / (1.0 + lambda);
    r.alpha2 = k.Omega0 * dt / (1.0 + lambda);
  }
  if (!stop && k.shear) {
    {
      double deltay = 1.5 * k.Omega0 * k.xlen * (t_cur + dt / 2);
      deltay = fmod(deltay, k.ylen);
      r.remap_jplus = (int)(deltay / k.dy);
      const double epsi = fmod(deltay, k.dy);
      r.remap_eps_min = 1.0 - epsi / k.dy;
      r.remap_eps_max = epsi / k.dy;
    }
    {
      double deltay = 1.5 * k.Omega0 * k.xlen * (t_cur + dt);
      deltay = fmod(deltay, k.ylen);
      r.ghost_jplus = (int)(deltay / k.dy);
      const double epsi = fmod(deltay, k.dy);
      r.ghost_eps_min = 1.0 - epsi / k.dy;
      r.ghost_eps_max = epsi / k.dy;
    }
  }
  *out = r;
}

}  // namespace rgpu_dev
