// launchers.h -- kernel functors: one small struct per launch, holding the launch's arguments by value and
// calling the per-cell body.  rg_kernel<BLOCK, K_xxx> is the __global__ entry the profiler sees.
#pragma once
#include "kernels_bc.h"
#include "kernels_dissipative.h"
#include "kernels_hydro.h"
#include "kernels_mhd2d.h"
#include "kernels_mhd3d.h"
#include "step_clock_rec.h"   // StepClock: the time step as a device record (kernels with a `clk` member read it instead of their by-value arguments)

namespace rgpu_dev {

// Launch-time specialisation of the heavy kernels (SPEC template parameter of the functors below).  Solver choice,
// equation of state, rotating frame and gravity are the same for every cell of a launch; left as run-time tests they cut the solvers' straight-line code into ~85 basic
// blocks and keep every variant's registers allocated.  SPEC tells the optimiser what the host has checked
// (spec_matches), everything else still comes from DevParams: 35.9 -> 28.8 ms for the 512^3 MRI launch.
enum {
  SPEC_NONE = 0,
  SPEC_HLLD = 1,        // riemannSolver = hlld and magRiemannSolver = hlld
  SPEC_ISOTHERMAL = 2,  // cIso > 0
  SPEC_ADIABATIC = 4,   // cIso <= 0
  SPEC_ROTATING = 8,    // Omega0 > 0
  SPEC_INERTIAL = 16,   // Omega0 <= 0
  SPEC_NO_GRAVITY = 32,
  SPEC_SLOPE2 = 64,     // slope_type = 2 (and with it the capped face-field slope type)
  SPEC_HYDRO_APPROX = 128, SPEC_HYDRO_HLL = 256, SPEC_HYDRO_HLLC = 512,   // hydro riemannSolver
  SPEC_SLOPE1 = 1024    // slope_type = 1
};
template <int SPEC>
RG_DEVFN void spec_assume(const DevParams& g) {
  if (SPEC & SPEC_HLLD) { RG_ASSUME(g.riemannSolver == 3); RG_ASSUME(g.magRiemannSolver == 0); }
  if (SPEC & SPEC_ISOTHERMAL) RG_ASSUME(g.cIso > 0);
  if (SPEC & SPEC_ADIABATIC) RG_ASSUME(!(g.cIso > 0));
  if (SPEC & SPEC_ROTATING) { RG_ASSUME(g.rot == 1); RG_ASSUME(g.Omega0 > 0); }
  if (SPEC & SPEC_INERTIAL) { RG_ASSUME(g.rot == 0); RG_ASSUME(!(g.Omega0 > 0)); }
  if (SPEC & SPEC_NO_GRAVITY) RG_ASSUME(g.grav_on == 0);
  if (SPEC & SPEC_SLOPE2) { RG_ASSUME(g.slope_type == 2.0); RG_ASSUME(g.mag_slope_type == 2.0); }
  if (SPEC & SPEC_HYDRO_APPROX) RG_ASSUME(g.riemannSolver == 0);
  if (SPEC & SPEC_HYDRO_HLL) RG_ASSUME(g.riemannSolver == 1);
  if (SPEC & SPEC_HYDRO_HLLC) RG_ASSUME(g.riemannSolver == 2);
  if (SPEC & SPEC_SLOPE1) RG_ASSUME(g.slope_type == 1.0);
}
inline bool spec_matches(int spec, const DevParams& g) {
  if ((spec & SPEC_HLLD) && !(g.riemannSolver == 3 && g.magRiemannSolver == 0)) return false;
  if ((spec & SPEC_ISOTHERMAL) && !(g.cIso > 0)) return false;
  if ((spec & SPEC_ADIABATIC) && (g.cIso > 0)) return false;
  if ((spec & SPEC_ROTATING) && !(g.rot == 1 && g.Omega0 > 0)) return false;
  if ((spec & SPEC_INERTIAL) && !(g.rot == 0 && !(g.Omega0 > 0))) return false;
  if ((spec & SPEC_NO_GRAVITY) && g.grav_on != 0) return false;
  if ((spec & SPEC_SLOPE2) && !(g.slope_type == 2.0 && g.mag_slope_type == 2.0)) return false;
  if ((spec & SPEC_HYDRO_APPROX) && g.riemannSolver != 0) return false;
  if ((spec & SPEC_HYDRO_HLL) && g.riemannSolver != 1) return false;
  if ((spec & SPEC_HYDRO_HLLC) && g.riemannSolver != 2) return false;
  if ((spec & SPEC_SLOPE1) && !(g.slope_type == 1.0)) return false;
  return true;
}


// ---- hydro -----------------------------------------------------------------------------------------------------
template <int NV>
struct K_hydro_prim {
  DevParams g; const double* U; double* Q;
  RG_DEVFN void operator()(unsigned idx) const { hydro_prim_cell<NV>(g, U, Q, idx); }
};
template <int ND, int NV, int SPEC = SPEC_NONE>
struct K_hydro_trace {
  DevParams g; const double* Q; double* T; double dtdx, dtdy, dtdz;
  RG_DEVFN void operator()(unsigned idx) const { spec_assume<SPEC>(g); hydro_trace_cell<ND, NV>(g, Q, T, dtdx, dtdy, dtdz, idx); }
};
// GF: per-cell gravity field (DevParams::G) instead of the uniform vector, see half_dt_gravity
template <int ND, int NV, bool GF = false, int SPEC = SPEC_NONE>
struct K_hydro_flux {
  DevParams g; const double* T; double* F;
  RG_DEVFN void operator()(unsigned idx) const { spec_assume<SPEC>(g); hydro_flux_cell<ND, NV, GF>(g, T, F, idx); }
};
template <int ND, int NV, bool GF = false>
struct K_hydro_update {
  DevParams g; const double* Uold; double* Unew; const double* F; double dtdx, dtdy, dtdz; unsigned long long* dt_slots;
  RG_DEVFN void operator()(unsigned idx) const { hydro_update_cell<ND, NV, GF>(g, Uold, Unew, F, dtdx, dtdy, dtdz, idx, dt_slots); }
};
template <int NV>
struct K_hydro_invdt {
  DevParams g; const double* U;
  RG_DEVFN double operator()(unsigned idx) const { return hydro_invdt_cell<NV>(g, U, idx); }
};

// ---- MHD -------------------------------------------------------------------------------------------------------
template <int SPEC = SPEC_NONE>
struct K_mhd_prim {
  DevParams g; const double* U; double* Q; double dt;
  RG_DEVFN void operator()(unsigned idx) const { spec_assume<SPEC>(g); mhd_prim_cell(g, U, Q, dt, idx); }
};
template <int SPEC = SPEC_NONE>
struct K_mhd_invdt {
  DevParams g; const double* U;
  RG_DEVFN double operator()(unsigned idx) const { spec_assume<SPEC>(g); return mhd_invdt_cell(g, U, idx); }
};
template <int ND>
struct K_visc_flux {
  DevParams g; const double* U; double* Fd; double nu, dt;
  RG_DEVFN void operator()(unsigned idx) const { visc_flux_cell<ND>(g, U, Fd, nu, dt, idx); }
};
template <int ND>
struct K_flux_update {
  DevParams g; double* U; const double* Fd; int v0, v1;
  RG_DEVFN void operator()(unsigned idx) const { flux_update_cell<ND>(g, U, Fd, v0, v1, idx); }
};
template <int ND>
struct K_resist_emf {
  DevParams g; const double* U; double* E; double eta;
  RG_DEVFN void operator()(unsigned idx) const { resist_emf_cell<ND>(g, U, E, eta, idx); }
};
template <int ND>
struct K_resist_ct {
  DevParams g; double* U; const double* E; double dtdx, dtdy, dtdz;
  RG_DEVFN void operator()(unsigned idx) const { resist_ct_cell<ND>(g, U, E, dtdx, dtdy, dtdz, idx); }
};
template <int ND>
struct K_resist_eflux {
  DevParams g; const double* U; double* Fd; double eta, dt;
  RG_DEVFN void operator()(unsigned idx) const { resist_eflux_cell<ND>(g, U, Fd, eta, dt, idx); }
};
struct K_hist_rows {
  DevParams g; const double* U; double* rows;
  RG_DEVFN void operator()(unsigned idx) const { hist_row_cell(g, U, rows, idx); }
};
struct K_hist_turb_rows {
  DevParams g; const double* U; double* rows;
  RG_DEVFN void operator()(unsigned idx) const { hist_turb_row_cell(g, U, rows, idx); }
};
struct K_hist_cols {
  DevParams g; const double* rows; double* cols; int nq;
  RG_DEVFN void operator()(unsigned idx) const { hist_col_cell(g, rows, cols, nq, idx); }
};
struct K_checksum_rows {
  DevParams g; const double* U; unsigned long long* rows;
  RG_DEVFN void operator()(unsigned idx) const { checksum_row_cell(g, U, rows, idx); }
};
struct K_hist_reynolds {
  DevParams g; const double* U; const double* mean_vx; const double* mean_vy; double dTau; double* rows;
  RG_DEVFN void operator()(unsigned idx) const { hist_reynolds_cell(g, U, mean_vx, mean_vy, dTau, rows, idx); }
};
struct K_mhd_trace2d {
  DevParams g; const double* U; const double* Q; double* T; double dtdx, dtdy;
  RG_DEVFN void operator()(unsigned idx) const { mhd_trace2d_cell(g, U, Q, T, dtdx, dtdy, idx); }
};
template <bool GF = false>
struct K_mhd_flux2d {
  DevParams g; const double* T; double* F;
  RG_DEVFN void operator()(unsigned idx) const { mhd_flux2d_cell<GF>(g, T, F, idx); }
};
template <bool GF = false>
struct K_mhd_update2d {
  DevParams g; RotCoef rc; const double* Uold; double* Unew; const double* F; double dt, dtdx, dtdy; unsigned long long* dt_slots;
  RG_DEVFN void operator()(unsigned idx) const { mhd_update2d_cell<GF>(g, rc, Uold, Unew, F, dt, dtdx, dtdy, idx, dt_slots); }
};
template <int SPEC = SPEC_NONE>
struct K_mhd_elec {
  DevParams g; const double* U; const double* Q; double* E;
  RG_DEVFN void operator()(unsigned idx) const { spec_assume<SPEC>(g); mhd_elec_cell(g, U, Q, E, idx); }
};
template <int SPEC = SPEC_NONE>
struct K_mhd_trace3d {
  DevParams g; const double* U; const double* Q; const double* E; double* T; double dtdx, dtdy, dtdz;
  RG_DEVFN void operator()(unsigned idx) const { spec_assume<SPEC>(g); mhd_trace3d_cell(g, U, Q, E, T, dtdx, dtdy, dtdz, idx); }
};
template <int MASK, bool GF = false, int SPEC = SPEC_NONE>
struct K_mhd_flux3d {
  DevParams g; const double* T; double* F; double* emf;
  RG_DEVFN void operator()(unsigned idx) const {
    spec_assume<SPEC>(g);
    mhd_flux3d_cell<MASK, GF>(g, T, F, emf, idx);
  }
};
struct K_forcing_rows {
  DevParams g; const double* U; const double* Frc; double* rows;
  RG_DEVFN void operator()(unsigned idx) const { forcing_row_cell(g, U, Frc, rows, idx); }
};
struct K_add_forcing {
  DevParams g; double* U; const double* Frc; double norm;
  RG_DEVFN void operator()(unsigned idx) const { add_forcing_cell(g, U, Frc, norm, idx); }
};
struct K_ou_forcing {
  DevParams g; double* U; rgpu_ou::OuModes M; double dt, yMin, zMin; int kz0;
  RG_DEVFN void operator()(unsigned idx) const { ou_forcing_cell(g, U, M, dt, yMin, zMin, kz0, idx); }
};
// (the ghost-fill functors end in `const StepClock* clk`: inside a batch of device-clock steps a stopped step must leave the state as it
//  is -- they take nothing else from the record.  Omitted in an aggregate initialiser it is 0.)
struct K_bc_zstrat {
  DevParams g; ZStrat zs; double* U; int side; const StepClock* clk;
  RG_DEVFN void operator()(unsigned ij) const { if (clk && clk->stop) return; zstrat_column(g, zs, U, side, ij); }
};
struct K_shear_save_emf {
  DevParams g; const double* emf; double* save;
  RG_DEVFN void operator()(unsigned idx) const { shear_save_emf_cell(g, emf, save, idx); }
};
struct K_shear_remap {
  DevParams g; ShearRemap sr; const double* F; double* emf; const double* save; double* remap; double dtdx; const StepClock* clk;
  RG_DEVFN void operator()(unsigned idx) const {
    if (clk) {   // device-side time step: offsets at t + dt/2 and dt/dx from the record
      if (clk->stop) return;
      const ShearRemap r = {clk->remap_jplus, clk->remap_eps_min, clk->remap_eps_max};
      shear_remap_cell(g, r, F, emf, save, remap, clk->dtdx, idx);
      return;
    }
    shear_remap_cell(g, sr, F, emf, save, remap, dtdx, idx);
  }
};
// a functor over two disjoint index ranges in one launch: t < n1 -> k(first1 + t), else k(first2 + t - n1)
template <class K>
struct K_two_ranges {
  K k; unsigned first1, n1, first2;
  RG_DEVFN void operator()(unsigned t) const { k(t < n1 ? first1 + t : first2 + (t - n1)); }
};
// t = z segment * (isize * jsize) + column: one thread marches planes [k_lo + seg * seg_len, + seg_len) of [k_lo, k_hi)
template <bool ROT, bool GF = false, int SPEC = SPEC_NONE>
struct K_mhd_update3d {
  DevParams g; RotCoef rc; const double* Uold; double* Unew; const double* F; const double* emf; const double* remap;
  double dt, dtdx, dtdy, dtdz; unsigned long long* dt_slots; int k_lo, k_hi, seg_len;
  // a second plane range in the same launch (the two boundary ranges of a slab): threads [n1, ...) march [k_lo2, k_hi2); n1 = 0xffffffff: none
  unsigned n1; int k_lo2, k_hi2;
  const StepClock* clk;   // device-side time step (0: the by-value dt, rc above)
  RG_DEVFN void operator()(unsigned t) const {
    spec_assume<SPEC>(g);
#ifdef RG_UPD_SETPRIO
    __builtin_amdgcn_s_setprio(RG_UPD_SETPRIO);
#endif
    const bool second = t >= n1;
    const unsigned tt = second ? t - n1 : t;
    const int lo = second ? k_lo2 : k_lo, hi = second ? k_hi2 : k_hi;
    // (one call site: launch-uniform selects of the scalars, not two copies of the column march)
    const bool dev = clk != 0;
    if (dev && clk->stop) return;
    const RotCoef r = {dev ? clk->lambda : rc.lambda, dev ? clk->ratio : rc.ratio, dev ? clk->alpha1 : rc.alpha1, dev ? clk->alpha2 : rc.alpha2};
    mhd_update3d_column<ROT, GF>(g, r, Uold, Unew, F, emf, remap, dev ? clk->dt : dt, dev ? clk->dtdx : dtdx, dev ? clk->dtdy : dtdy, dev ? clk->dtdz : dtdz,
                                 tt, lo, hi, seg_len, dt_slots);
  }
};

// ---- boundaries -------------------------------------------------------------------------------------------------
struct K_bc_face {
  DevParams g; double* U; int dir, side, bct; const StepClock* clk;
  RG_DEVFN void operator()(unsigned idx) const { if (clk && clk->stop) return; bc_face_cell(g, U, dir, side, bct, idx); }
};
// both faces of one direction in one launch (they read interior cells only, so they do not depend on each other): the first
// n indices are the low face, the next n the high face
struct K_bc_faces {
  DevParams g; double* U; int dir, bct_lo, bct_hi; unsigned n; const StepClock* clk;
  RG_DEVFN void operator()(unsigned idx) const {
    if (clk && clk->stop) return;
    if (idx < n) bc_face_cell(g, U, dir, 0, bct_lo, idx);
    else bc_face_cell(g, U, dir, 1, bct_hi, idx - n);
  }
};
struct K_bc_faces_range {   // ... restricted to the face indices [first, first + n) of each face (a range of z planes)
  K_bc_faces f; unsigned first;
  RG_DEVFN void operator()(unsigned idx) const {
    if (f.clk && f.clk->stop) return;
    if (idx < f.n) bc_face_cell(f.g, f.U, f.dir, 0, f.bct_lo, first + idx);
    else bc_face_cell(f.g, f.U, f.dir, 1, f.bct_hi, first + (idx - f.n));
  }
};
struct K_jet {
  DevParams g; JetParams jp; double* U; const StepClock* clk;
  RG_DEVFN void operator()(unsigned idx) const { if (clk && clk->stop) return; jet_cell(g, jp, U, idx); }
};
// the in-plane ghost fill in one pass (fill_xy_cell) over the planes [lo1, lo1 + n1) and [lo2, ...): idx = plane * per_plane + ghost cell
struct K_fill_xy {
  DevParams g; FillXY f; double* U; unsigned per_plane; int lo1, n1, lo2;
  const StepClock* clk;   // device-side time step: the shearing-box offsets at t + dt come from the record (a stopped step fills nothing)
  RG_DEVFN void operator()(unsigned idx) const {
    const unsigned p = idx / per_plane, t = idx - p * per_plane;
    const int k = (int)p < n1 ? lo1 + (int)p : lo2 + ((int)p - n1);
    if (clk) {
      if (clk->stop) return;
      FillXY fc = f;
      fc.sg.jplus = clk->ghost_jplus; fc.sg.eps_min = clk->ghost_eps_min; fc.sg.eps_max = clk->ghost_eps_max;
      fill_xy_cell(g, fc, U, t, k);
      return;
    }
    fill_xy_cell(g, f, U, t, k);
  }
};
struct K_shear_ghost {
  DevParams g; ShearGhost sg; double* U;
  RG_DEVFN void operator()(unsigned idx) const { shear_ghost_cell(g, sg, U, idx); }
};

}  // namespace rgpu_dev
