// orc_hydro.cpp -- ORACLE (test infrastructure).  Hydro unsplit step, "unsplitVersion 1".
//   godunov_unsplit_cpu      HydroRunGodunov.cpp:1820-1875  (ghost fill of the input, copy, primitives)
//   convertToPrimitives      HydroRunGodunov.cpp:4133-4262
//   godunov_unsplit_cpu_v1   HydroRunGodunov.cpp:2437-2949  (trace over [1,size-1), flux + scatter update)
#include "orc_pointwise.h"

namespace orc {

namespace {

template <int NDIM, int NV>
void hydro_step_t(const Ctx& c, double* Uold_d, double* Unew_d, double dt) {
  const rgpu_params& p = c.p;
  const int gw = c.gw, isize = c.isize, jsize = c.jsize, ksize = c.ksize;
  const double dtdx = dt / c.dx, dtdy = dt / c.dy, dtdz = dt / c.dz;
  const size_t N = c.ncell;

  make_all_boundaries(c, Uold_d, 0.0, 0.0);
  std::memcpy(Unew_d, Uold_d, sizeof(double) * N * NV);

  Field Unew; Unew.wrap(c, Unew_d, NV);
  Field Q, qm[3], qp[3];
  Q.alloc(c, NV);
  for (int d = 0; d < NDIM; ++d) { qm[d].alloc(c, NV); qp[d].alloc(c, NV); }

  // primitive variables over the WHOLE array (ghosts included)
  for (size_t o = 0; o < N; ++o) {
    double u[NV], q[NV], cs;
    for (int v = 0; v < NV; ++v) u[v] = Uold_d[o + v * N];
    hydro_constoprim<NV>(p, u, q, cs);
    for (int v = 0; v < NV; ++v) Q.d[o + v * N] = q[v];
  }

  const int k0 = (NDIM == 3) ? 1 : 0, k1 = (NDIM == 3) ? ksize - 1 : 1;
  const size_t stride[3] = {1, (size_t)isize, (size_t)isize * jsize};

  // slopes + trace
  for (int k = k0; k < k1; k++)
    for (int j = 1; j < jsize - 1; j++)
      for (int i = 1; i < isize - 1; i++) {
        const size_t o = c.idx(i, j, k);
        double q[NV], dq[NDIM][NV], tqm[NDIM][NV], tqp[NDIM][NV];
        for (int v = 0; v < NV; ++v) q[v] = Q.d[o + v * N];
        for (int d = 0; d < NDIM; ++d)
          for (int v = 0; v < NV; ++v) {
            const double qP = Q.d[o + stride[d] + v * N], qM = Q.d[o - stride[d] + v * N];
            if (p.slope_type == 0) dq[d][v] = 0.0;
            else if (NDIM == 3 && p.slope_type == 1) dq[d][v] = minmod_slope(qM, q[v], qP);
            else dq[d][v] = tvd_slope(p.slope_type, qM, q[v], qP);
          }
        hydro_trace<NDIM, NV>(p, q, dq, dtdx, dtdy, dtdz, tqm, tqp);
        if (p.gravityEnabled) {   // gravity predictor on every traced face state (HydroRunGodunov.cpp:2485-2497, 2705-2734)
          for (int d = 0; d < NDIM; ++d)
            for (int e = 0; e < NDIM; ++e) {
              tqm[d][IU + e] += 0.5 * dt * c.grav(i, j, k, e);
              tqp[d][IU + e] += 0.5 * dt * c.grav(i, j, k, e);
            }
        }
        for (int d = 0; d < NDIM; ++d)
          for (int v = 0; v < NV; ++v) { qm[d].d[o + v * N] = tqm[d][v]; qp[d].d[o + v * N] = tqp[d][v]; }
      }

  // Riemann problems at the low faces of every cell of [gw, size-gw] + scatter update.
  // unsplitVersion 1 (godunov_unsplit_cpu_v1, HydroRunGodunov.cpp:2437-2949): one sweep, the x, y, z fluxes of a cell
  // applied together.  unsplitVersion 2 (godunov_unsplit_cpu_v2, :2955-3849): one sweep per direction -- the same
  // fluxes (its direction-wise trace evaluates the same expressions), but a cell receives them in the order
  // +Fx, -Fx', +Fy, -Fy', +Fz, -Fz' instead of +Fx, +Fy, +Fz, -Fx', -Fy', -Fz'.
  const int kb0 = (NDIM == 3) ? gw : 0, kb1 = (NDIM == 3) ? ksize - gw + 1 : 1;
  const int nsweep = (p.unsplitVersion == 2) ? NDIM : 1;
  for (int sweep = 0; sweep < nsweep; ++sweep)
  for (int k = kb0; k < kb1; k++)
    for (int j = gw; j < jsize - gw + 1; j++)
      for (int i = gw; i < isize - gw + 1; i++) {
        const size_t o = c.idx(i, j, k);
        double ql[NV], qr[NV], flux[3][NV];
        // face-normal frame: swap IU with the normal velocity
        for (int d = 0; d < NDIM; ++d) {
          if (nsweep > 1 && d != sweep) continue;
          const int swp = (d == 0) ? IU : (d == 1) ? IV : IW;
          for (int v = 0; v < NV; ++v) {
            int vs = v;
            if (v == IU) vs = swp; else if (v == swp) vs = IU;
            ql[v] = qm[d].d[o - stride[d] + vs * N];
            qr[v] = qp[d].d[o + vs * N];
          }
          for (int v = 0; v < NV; ++v) flux[d][v] = 0.0;  // the reference leaves flux untouched for unknown solvers
          hydro_riemann<NV>(p, ql, qr, flux[d]);
        }
        const bool in_i = i < isize - gw, in_j = j < jsize - gw, in_k = (NDIM == 3) ? (k < ksize - gw) : true;
        const bool do_x = nsweep == 1 || sweep == 0, do_y = nsweep == 1 || sweep == 1, do_z = NDIM == 3 && (nsweep == 1 || sweep == 2);
        // x
        if (do_x && i > gw && in_j && in_k)
          for (int v = 0; v < NV; ++v) Unew.d[o - 1 + v * N] -= flux[0][v] * dtdx;
        if (do_x && in_i && in_j && in_k)
          for (int v = 0; v < NV; ++v) Unew.d[o + v * N] += flux[0][v] * dtdx;
        // y (IU <-> IV swapped back)
        if (do_y && in_i && j > gw && in_k)
          for (int v = 0; v < NV; ++v) {
            const int vs = (v == IU) ? IV : (v == IV) ? IU : v;
            Unew.d[o - stride[1] + v * N] -= flux[1][vs] * dtdy;
          }
        if (do_y && in_i && in_j && in_k)
          for (int v = 0; v < NV; ++v) {
            const int vs = (v == IU) ? IV : (v == IV) ? IU : v;
            Unew.d[o + v * N] += flux[1][vs] * dtdy;
          }
        if (do_z) {
          if (in_i && in_j && k > gw)
            for (int v = 0; v < NV; ++v) {
              const int vs = (v == IU) ? IW : (v == IW) ? IU : v;
              Unew.d[o - stride[2] + v * N] -= flux[2][vs] * dtdz;
            }
          if (in_i && in_j && in_k)
            for (int v = 0; v < NV; ++v) {
              const int vs = (v == IU) ? IW : (v == IW) ? IU : v;
              Unew.d[o + v * N] += flux[2][vs] * dtdz;
            }
        }
      }

  // gravity source term on the momenta of the interior (compute_gravity_source_term, HydroRunBase.cpp:1925-1985);
  // the total energy is left alone, as in the reference
  if (p.gravityEnabled) {
    const int kg0 = (NDIM == 3) ? gw : 0, kg1 = (NDIM == 3) ? ksize - gw : 1;
    for (int k = kg0; k < kg1; k++)
      for (int j = gw; j < jsize - gw; j++)
        for (int i = gw; i < isize - gw; i++) {
          const size_t o = c.idx(i, j, k);
          const double rhoOld = Uold_d[o + ID * N], rhoNew = Unew.d[o + ID * N];
          for (int e = 0; e < NDIM; ++e) Unew.d[o + (IU + e) * N] += 0.5 * dt * c.grav(i, j, k, e) * (rhoOld + rhoNew);
        }
  }
  dissipative_stage(c, Unew_d, dt, 0.0);   // [hydro] nu > 0 (HydroRunGodunov.cpp:2620-2640, 2908-2925)
  random_forcing(c, Unew_d, dt);           // problem "turbulence" (HydroRunGodunov.cpp:2930-2938, 3830-3838)
  ou_forcing(c, Unew_d, dt);               // problem "turbulence-Ornstein-Uhlenbeck" (:2940-2945)
}

}  // namespace

void hydro_step(const Ctx& c, double* Uold, double* Unew, double dt) {
  if (c.three_d) hydro_step_t<3, 5>(c, Uold, Unew, dt);
  else hydro_step_t<2, 4>(c, Uold, Unew, dt);
}

}  // namespace orc
