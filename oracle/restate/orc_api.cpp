// orc_api.cpp -- ORACLE (test infrastructure).  C entry points of the CPU restatement, loaded with ctypes by
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg ONLY.  The product never links this.
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <vector>

#include "orc_common.h"

using namespace orc;

extern "C" {

// make_boundaries(U, idim)
int orc_make_boundaries(const rgpu_params* p, double* U, int idim) {
  Ctx c(*p);
  make_boundaries(c, U, idim);
  return 0;
}

// make_all_boundaries / make_all_boundaries_shear
int orc_make_all_boundaries(const rgpu_params* p, double* U, double totalTime, double dt) {
  Ctx c(*p);
  make_all_boundaries(c, U, totalTime, dt);
  return 0;
}

int orc_make_boundaries_shear(const rgpu_params* p, double* U, double totalTime, double dt) {
  Ctx c(*p);
  make_boundaries_shear(c, U, totalTime, dt);
  return 0;
}

double orc_compute_inv_dt(const rgpu_params* p, const double* U) {
  Ctx c(*p);
  return compute_inv_dt(c, U);
}

// compute_dt[_mhd]
double orc_compute_dt(const rgpu_params* p, const double* U) {
  Ctx c(*p);
  return p->cfl / compute_inv_dt(c, U);
}

// history_mri (MHDRunBase.cpp:3476-3619; history_default :3311-3407 prints its mass and divB): the reference's loops,
// in its order.  out[8] = mass, maxwell, reynolds, magp, mean_Bx, mean_By, mean_Bz, divB
void orc_history_mri(const rgpu_params* p, const double* U, double* out) {
  Ctx c(*p);
  const int gw = c.gw, isize = c.isize, jsize = c.jsize, ksize = c.ksize;
  const size_t N = c.ncell;
  const bool three_d = c.three_d;
  const int k0 = three_d ? gw : 0, k1 = three_d ? ksize - gw : 1;
  const size_t sj = (size_t)isize, sk = three_d ? (size_t)isize * jsize : 0;
  const double dx = c.dx, dy = c.dy, dz = c.dz;
  double mass = 0.0, magp = 0.0, maxwell = 0.0, mean_Bx = 0.0, mean_By = 0.0, mean_Bz = 0.0;
  for (int k = k0; k < k1; k++)
    for (int j = gw; j < jsize - gw; j++)
      for (int i = gw; i < isize - gw; i++) {
        const size_t o = c.idx(i, j, k);
        mass += U[o + ID * N];
        magp += 0.25 * ((U[o + IA * N] + U[o + 1 + IA * N]) * (U[o + IA * N] + U[o + 1 + IA * N]));
        magp += 0.25 * ((U[o + IB * N] + U[o + sj + IB * N]) * (U[o + IB * N] + U[o + sj + IB * N]));
        if (three_d) magp += 0.25 * ((U[o + IC * N] + U[o + sk + IC * N]) * (U[o + IC * N] + U[o + sk + IC * N]));
        maxwell -= 0.25 * (U[o + IA * N] + U[o + 1 + IA * N]) * (U[o + IB * N] + U[o + sj + IB * N]);
        mean_Bx += U[o + IA * N]; mean_By += U[o + IB * N]; mean_Bz += U[o + IC * N];
      }
  double dTau;
  if (three_d) dTau = dx * dy * dz / (p->xMax - p->xMin) / (p->yMax - p->yMin) / (p->zMax - p->zMin);
  else dTau = dx * dy / (p->xMax - p->xMin) / (p->yMax - p->yMin);
  magp = magp * dTau / 2.; mass = mass * dTau; maxwell = maxwell * dTau;
  mean_Bx = mean_Bx * dTau; mean_By = mean_By * dTau; mean_Bz = mean_Bz * dTau;
  std::vector<double> lm((size_t)isize * 3, 0.0);
  for (int k = k0; k < k1; k++)
    for (int j = gw; j < jsize - gw; j++)
      for (int i = 0; i < isize; i++) {
        const size_t o = c.idx(i, j, k);
        lm[i] += U[o + ID * N];
        lm[isize + i] += U[o + IU * N] / U[o + ID * N];
        lm[2 * isize + i] += U[o + IV * N] / U[o + ID * N];
      }
  const int nyz = p->ny * (three_d ? p->nz : 1);
  for (int i = 0; i < 3 * isize; i++) lm[i] /= nyz;
  double reynolds = 0.0, divB = 0.0;
  for (int k = k0; k < k1; k++)
    for (int j = gw; j < jsize - gw; j++)
      for (int i = gw; i < isize - gw; i++) {
        const size_t o = c.idx(i, j, k);
        reynolds += U[o + ID * N] * dTau * (U[o + IU * N] / U[o + ID * N] - lm[isize + i]) * (U[o + IV * N] / U[o + ID * N] - lm[2 * isize + i]);
        double dv = (U[o + 1 + IA * N] - U[o + IA * N]) / dx + (U[o + sj + IB * N] - U[o + IB * N]) / dy;
        if (three_d) dv = dv + (U[o + sk + IC * N] - U[o + IC * N]) / dz;
        divB += dv;
      }
  out[0] = mass; out[1] = maxwell; out[2] = reynolds; out[3] = magp; out[4] = mean_Bx; out[5] = mean_By; out[6] = mean_Bz; out[7] = divB;
}

// history_turbulence (MHDRunBase.cpp:3626-3810), 3D: the reference's two loops in its order.  out[18] = the columns after
// totalTime and dt: mass divB eKin eMag helicity mean_rho mean_B mean_Bx mean_By mean_Bz mean_rhovx mean_rhovy mean_rhovz
// Ma_s Ma_alfven coef_x coef_y coef_z
void orc_history_turbulence(const rgpu_params* p, const double* U, double* out) {
  Ctx c(*p);
  const int ghostWidth = c.gw, isize = c.isize, jsize = c.jsize, ksize = c.ksize;
  const int nx = p->nx, ny = p->ny, nz = p->nz;
  const size_t N = c.ncell;
  const double dx = p->dx, dy = p->dy, dz = p->dz;
  const double pi = 2 * asin(1.0);
  double mass = 0.0, eKin = 0.0, eMag = 0.0;
  double helicity = 0.0;
  double mean_Bx = 0.0, mean_By = 0.0, mean_Bz = 0.0;
  double mean_rhovx = 0.0, mean_rhovy = 0.0, mean_rhovz = 0.0;
  double mean_v2 = 0.0, mean_rho = 0.0;
  int kfft = nx - 3;
  double coef_x_re = 0.0, coef_x_im = 0.0, coef_y_re = 0.0, coef_y_im = 0.0, coef_z_re = 0.0, coef_z_im = 0.0;
#define SQR(x) ((x) * (x))
  for (int k = ghostWidth; k < ksize - ghostWidth; k++)
    for (int j = ghostWidth; j < jsize - ghostWidth; j++)
      for (int i = ghostWidth; i < isize - ghostWidth; i++) {
        const size_t o = c.idx(i, j, k);
        double rho = U[o + ID * N];
        double bx = U[o + IA * N];
        mass += rho;
        eKin += SQR(U[o + IU * N]) / rho;
        eKin += SQR(U[o + IV * N]) / rho;
        eKin += SQR(U[o + IW * N]) / rho;
        mean_v2 += SQR(U[o + IU * N] / rho);
        mean_v2 += SQR(U[o + IV * N] / rho);
        mean_v2 += SQR(U[o + IW * N] / rho);
        eMag += SQR(U[o + IA * N]);
        eMag += SQR(U[o + IB * N]);
        eMag += SQR(U[o + IC * N]);
        helicity += U[o + IU * N] * U[o + IA * N] / sqrt(rho);
        helicity += U[o + IV * N] * U[o + IB * N] / sqrt(rho);
        helicity += U[o + IW * N] * U[o + IC * N] / sqrt(rho);
        mean_Bx += U[o + IA * N];
        mean_By += U[o + IB * N];
        mean_Bz += U[o + IC * N];
        mean_rhovx += U[o + IU * N];
        mean_rhovy += U[o + IV * N];
        mean_rhovz += U[o + IW * N];
        mean_rho += rho;
        coef_x_re += bx * cos(2 * pi * kfft * i / nx);
        coef_x_im += bx * sin(2 * pi * kfft * i / nx);
        coef_y_re += bx * cos(2 * pi * kfft * j / ny);
        coef_y_im += bx * sin(2 * pi * kfft * j / ny);
        coef_z_re += bx * cos(2 * pi * kfft * k / nz);
        coef_z_im += bx * sin(2 * pi * kfft * k / nz);
      }
  double dTau = dx * dy * dz / (p->xMax - p->xMin) / (p->yMax - p->yMin) / (p->zMax - p->zMin);
  mass = mass * dTau;
  eKin = eKin * dTau;
  eMag = eMag * dTau;
  helicity *= dTau;
  mean_Bx = mean_Bx * dTau; mean_By = mean_By * dTau; mean_Bz = mean_Bz * dTau;
  double mean_B = sqrt(SQR(mean_Bx) + SQR(mean_By) + SQR(mean_Bz));
  mean_rhovx = mean_rhovx * dTau; mean_rhovy = mean_rhovy * dTau; mean_rhovz = mean_rhovz * dTau;
  mean_v2 = mean_v2 * dTau;
  mean_rho = mean_rho * dTau;
  double coef_x = sqrt(SQR(coef_x_re) + SQR(coef_x_im)); coef_x *= dTau;
  double coef_y = sqrt(SQR(coef_y_re) + SQR(coef_y_im)); coef_y *= dTau;
  double coef_z = sqrt(SQR(coef_z_re) + SQR(coef_z_im)); coef_z *= dTau;
  double divB = 0.0;
  for (int k = ghostWidth; k < ksize - ghostWidth; k++)
    for (int j = ghostWidth; j < jsize - ghostWidth; j++)
      for (int i = ghostWidth; i < isize - ghostWidth; i++) {
        const size_t o = c.idx(i, j, k);
        divB += (U[c.idx(i + 1, j, k) + IA * N] - U[o + IA * N]) / dx + (U[c.idx(i, j + 1, k) + IB * N] - U[o + IB * N]) / dy +
                (U[c.idx(i, j, k + 1) + IC * N] - U[o + IC * N]) / dz;
      }
#undef SQR
  double Ma_alfven = sqrt(mean_v2) / (mean_B / sqrt(4 * pi * mean_rho));
  double Ma_s = sqrt(mean_v2) / p->cIso;
  const double v[18] = {mass, divB, eKin, eMag, helicity, mean_rho, mean_B, mean_Bx, mean_By, mean_Bz, mean_rhovx, mean_rhovy, mean_rhovz,
                        Ma_s, Ma_alfven, coef_x, coef_y, coef_z};
  for (int q = 0; q < 18; q++) out[q] = v[q];
}

static int check_scope(const rgpu_params* p) {
  if (p->slope_type != 0 && p->slope_type != 1 && p->slope_type != 2 && p->slope_type != 3) return RGPU_EUNSUPPORTED;
  // slope_type 3 (positivity preserving) exists in the 2D MHD and the plain 3D MHD steps only; the hydro steps and
  // the rotating 3D step call slope routines that leave dq unset for it (slope.h:97-147,324-427; slope_mhd.h:436-502)
  if (p->slope_type == 3 && (!p->mhdEnabled || (p->Omega0 > 0 && p->nz_global != 1))) return RGPU_EUNSUPPORTED;
  if (p->randomForcingEnabled && (p->nz_global == 1 || (p->mhdEnabled && p->Omega0 > 0))) return RGPU_EUNSUPPORTED;
  if (p->ouForcingEnabled && (p->nz_global == 1 || (p->mhdEnabled && p->Omega0 > 0))) return RGPU_EUNSUPPORTED;
  if (p->mhdEnabled) {
    const bool three_d = p->nz_global != 1;
    if (p->magRiemannSolver != RGPU_MAG_HLLD && p->magRiemannSolver != RGPU_MAG_HLLA && p->magRiemannSolver != RGPU_MAG_HLLF &&
        p->magRiemannSolver != RGPU_MAG_LLF) return RGPU_EUNSUPPORTED;
    if (!three_d && p->implementationVersion != 1 && p->implementationVersion != 0) return RGPU_EUNSUPPORTED;
    if (!three_d && p->Omega0 > 0 && p->shearingBoxEnabled) return RGPU_EUNSUPPORTED;   // "not fully implemented" in the reference
    if (three_d && !(p->Omega0 > 0) && p->implementationVersion != 3 && p->implementationVersion != 4) return RGPU_EUNSUPPORTED;
  } else {
    if (p->unsplitVersion != 1 && p->unsplitVersion != 2) return RGPU_EUNSUPPORTED;
  }
  return 0;
}

// h_gravity of the runs with gravityEnabled == 2: G[3][ksize][jsize][isize], caller-owned, used by every later call
// (0 = forget it)
void orc_set_gravity_field(const double* G) { Ctx::gravity_field() = G; }
// h_randomForcing of the "turbulence" problem, same convention
void orc_set_forcing_field(const double* F) { Ctx::forcing_field() = F; }

// godunov_unsplit(nStep, dt): Uold -> Unew (both ghost-inclusive, caller-owned)
int orc_godunov_unsplit(const rgpu_params* p, double* Uold, double* Unew, double dt, double totalTime) {
  const int rc = check_scope(p);
  if (rc) return rc;
  Ctx c(*p);
  ou_forget();   // a single step carries no forcing process (orc_run owns one)
  if (!p->mhdEnabled) hydro_step(c, Uold, Unew, dt);
  else if (!c.three_d) mhd_step_2d(c, Uold, Unew, dt);
  else mhd_step_3d(c, Uold, Unew, dt, totalTime);
  return 0;
}

// The time loop of start() (MHDRunGodunov.cpp:3801-3989 / HydroRunGodunov.cpp:3857-...): initial ghost fill,
// copy to U2, then oneStepIntegration until nStepmax / tEnd.  On return U holds the state of the LAST step
// (whatever its parity), *nsteps_done and *t_final are set and dts[0..nsteps_done) the time steps used.
int orc_run(const rgpu_params* p, double* U, int nStepmax, double tEnd, int* nsteps_done, double* t_final, double* dts) {
  const int rc = check_scope(p);
  if (rc) return rc;
  Ctx c(*p);
  const size_t n = c.ncell * c.nvar;
  std::vector<double> U2(n);
  make_all_boundaries(c, U, 0.0, 0.0);
  std::memcpy(U2.data(), U, sizeof(double) * n);
  ou_init(*p);   // init_forcing() of the initial condition (HydroRunBase.cpp:6990)
  double t = 0.0;
  int nStep = 0;
  while (t < tEnd && nStep < nStepmax) {
    double* cur = (nStep % 2 == 0) ? U : U2.data();
    double* nxt = (nStep % 2 == 0) ? U2.data() : U;
    const double dt = p->cfl / compute_inv_dt(c, cur);
    if (!p->mhdEnabled) hydro_step(c, cur, nxt, dt);
    else if (!c.three_d) mhd_step_2d(c, cur, nxt, dt);
    else mhd_step_3d(c, cur, nxt, dt, t);
    if (dts) dts[nStep] = dt;
    nStep++;
    t += dt;
  }
  ou_forget();
  if (nStep % 2 == 1) std::memcpy(U, U2.data(), sizeof(double) * n);
  if (nsteps_done) *nsteps_done = nStep;
  if (t_final) *t_final = t;
  return 0;
}

// orc_run with every loop nest of the 3D MHD step cut into z-slabs over `nthreads` threads (mhd_step_3d_mt): the all-cores
// CPU baseline of bench.py.  Same results as orc_run, bit for bit.  Scope: 3D MHD without gravity, dissipative stage, forcing,
// slope_type 3 (what the bench workloads use); anything else returns RGPU_EUNSUPPORTED.
// scan / nscan (may be 0): thread counts to try, one step each, at the start of the run -- the rest of the steps use the fastest
// (the step streams ~1.3 kB of intermediates per cell through 144 arrays: on a two-socket host the best count is well below
// the number of hardware threads).  step_seconds (may be 0): wall time of every step; *threads_used: the count the run settled on.
int orc_run_mt_scan(const rgpu_params* p, double* U, int nStepmax, double tEnd, int nthreads, const int* scan, int nscan, int* nsteps_done,
                    double* t_final, double* dts, double* step_seconds, int* threads_used) {
  const int rc = check_scope(p);
  if (rc) return rc;
  Ctx c(*p);
  if (!p->mhdEnabled || !c.three_d || p->gravityEnabled || p->nu > 0 || p->eta > 0 || p->randomForcingEnabled || p->ouForcingEnabled ||
      p->slope_type == 3)
    return RGPU_EUNSUPPORTED;
  const size_t n = c.ncell * c.nvar;
  make_all_boundaries(c, U, 0.0, 0.0);
  int most = nthreads;
  for (int i = 0; i < nscan; ++i) most = scan[i] > most ? scan[i] : most;
  // the two state arrays of the run: pages FIRST TOUCHED slab by slab by the threads that will work on them (the caller's array
  // lives wherever the caller's one thread put it: on a two-socket host every other slab would stream it across the link)
  struct Buf { double* p; Buf(size_t k) : p(static_cast<double*>(std::malloc(k * sizeof(double)))) {} ~Buf() { std::free(p); } };
  Buf Ua(n), Ub(n);
  if (!Ua.p || !Ub.p) return RGPU_ENOMEM;
  {
    const size_t plane = (size_t)c.isize * c.jsize, N = c.ncell;
    slabs(0, c.ksize, most, [&](int ka, int kb) {
      for (int v = 0; v < c.nvar; ++v) {
        std::memcpy(Ua.p + v * N + plane * ka, U + v * N + plane * ka, sizeof(double) * plane * (kb - ka));
        std::memcpy(Ub.p + v * N + plane * ka, U + v * N + plane * ka, sizeof(double) * plane * (kb - ka));
      }
    });
  }
  MtWork work(c, most);
  double t = 0.0, best = 1e300;
  int nStep = 0, use = nthreads;
  while (t < tEnd && nStep < nStepmax) {
    double* cur = (nStep % 2 == 0) ? Ua.p : Ub.p;
    double* nxt = (nStep % 2 == 0) ? Ub.p : Ua.p;
    const int nt = nStep < nscan ? scan[nStep] : use;
    const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    const double dt = p->cfl / compute_inv_dt_mhd3d_mt(c, cur, nt);
    mhd_step_3d_mt(c, work, cur, nxt, dt, t, nt);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (step_seconds) step_seconds[nStep] = secs;
    if (nStep < nscan && secs < best) { best = secs; use = nt; }
    if (dts) dts[nStep] = dt;
    nStep++;
    t += dt;
  }
  std::memcpy(U, (nStep % 2 == 1) ? Ub.p : Ua.p, sizeof(double) * n);
  if (nsteps_done) *nsteps_done = nStep;
  if (t_final) *t_final = t;
  if (threads_used) *threads_used = use;
  return 0;
}
// placement of the threads of orc_run_mt / orc_run_mt_scan: 0 = not pinned, 1 = pinned over all allowed CPUs in NUMA-node order,
// 2 = pinned inside the first NUMA node (one socket); returns the number of CPUs in the set (0: unpinned)
int orc_set_thread_placement(int mode) { return set_thread_placement(mode); }
int orc_run_mt(const rgpu_params* p, double* U, int nStepmax, double tEnd, int nthreads, int* nsteps_done, double* t_final, double* dts) {
  return orc_run_mt_scan(p, U, nStepmax, tEnd, nthreads, 0, 0, nsteps_done, t_final, dts, 0, 0);
}

}  // extern "C"
