// orc_dissipative.cpp -- ORACLE (test infrastructure).  The operator-split dissipative stage that follows the Godunov
// update when [hydro] nu > 0 or [MHD] eta > 0:
//   viscous fluxes + update      compute_viscosity_flux / compute_hydro_update   HydroRunBase.cpp:431-560, 876-1160, 1461-1533
//   resistive emf + CT update    compute_resistivity_emf_{2d,3d}, compute_ct_update MHDRunBase.cpp:256-344, 455-575
//   resistive energy flux        compute_resistivity_energy_flux_{2d,3d}          MHDRunBase.cpp:697-900
//                                + compute_hydro_update_energy                     HydroRunBase.cpp:1633-1700
// call sites: HydroRunGodunov.cpp:2620-2640, 2908-2925; mhd_godunov_unsplit_cpu_v1.cpp:244-272;
// mhd_godunov_unsplit_cpu_v3.cpp:662-694; MHDRunGodunov.cpp:3379-3420 (rotating).
// Written direction-generically (one loop body for the x, y and z faces); every expression keeps the reference's
// operand order.
#include <cmath>
#include <vector>

#include "orc_common.h"

namespace orc {

namespace {

inline double vel(const double* U, size_t N, size_t o, int k) { return U[o + (size_t)(IU + k) * N] / U[o + (size_t)ID * N]; }

// viscous fluxes F[d] (nvh components each) at the low faces of every cell of [gw, size-gw]
void viscosity_flux(const Ctx& c, const double* U, std::vector<double>* F, double dt) {
  const rgpu_params& p = c.p;
  const int ND = c.three_d ? 3 : 2, nvh = ND + 2;
  const size_t N = c.ncell;
  const size_t st[3] = {1, (size_t)c.isize, (size_t)c.isize * c.jsize};
  const double h[3] = {c.dx, c.dy, c.dz};
  const double two3rd = 2. / 3.;
  const double nu = p.nu, cIso = p.cIso;
  const int k0 = c.three_d ? c.gw : 0, k1 = c.three_d ? c.ksize - c.gw + 1 : 1;
  for (int k = k0; k < k1; k++)
    for (int j = c.gw; j < c.jsize - c.gw + 1; j++)
      for (int i = c.gw; i < c.isize - c.gw + 1; i++) {
        const size_t o = c.idx(i, j, k);
        for (int D = 0; D < ND; ++D) {
          const size_t oL = o - st[D];
          const double rho = 0.5 * (U[o + ID * N] + U[oL + ID * N]);
          double uavg[3] = {0, 0, 0};
          if (cIso <= 0) for (int a = 0; a < ND; ++a) uavg[a] = 0.5 * (vel(U, N, o, a) + vel(U, N, oL, a));
          double grad[3][3];  // grad[b][a] = d u_a / d x_b at the face
          for (int a = 0; a < ND; ++a) grad[D][a] = (vel(U, N, o, a) - vel(U, N, oL, a)) / h[D];
          for (int T = 0; T < ND; ++T) {
            if (T == D) continue;
            for (int a = 0; a < ND; ++a) {
              if (a != D && a != T) continue;
              const double uR = vel(U, N, o + st[T], a) + vel(U, N, oL + st[T], a);
              const double uL = vel(U, N, o - st[T], a) + vel(U, N, oL - st[T], a);
              grad[T][a] = (uR - uL) / h[T] / 4;
            }
          }
          double t[3];  // t[a] = stress component (D,a)
          {
            double tr = 2.0 * grad[D][D];
            for (int T = 0; T < ND; ++T) if (T != D) tr = tr - grad[T][T];
            t[D] = -two3rd * nu * rho * tr;
          }
          for (int T = 0; T < ND; ++T) {
            if (T == D) continue;
            const int a = (D < T) ? D : T, b = (D < T) ? T : D;
            t[T] = -nu * rho * (grad[b][a] + grad[a][b]);
          }
          double* f = F[D].data();
          f[o + ID * N] = 0.0;
          for (int a = 0; a < ND; ++a) f[o + (size_t)(IU + a) * N] = t[a] * dt / h[D];
          if (cIso <= 0) {
            double e = uavg[0] * t[0];
            for (int a = 1; a < ND; ++a) e = e + uavg[a] * t[a];
            f[o + IP * N] = e * dt / h[D];
          } else {
            f[o + IP * N] = 0.0;
          }
          (void)nvh;
        }
      }
}

// U(v) += (Fx(i) - Fx(i+1)); += (Fy(j) - Fy(j+1)); += (Fz(k) - Fz(k+1)) over the interior, variables [v0, v1)
void flux_update(const Ctx& c, double* U, const std::vector<double>* F, int v0, int v1) {
  const int ND = c.three_d ? 3 : 2;
  const size_t N = c.ncell;
  const size_t st[3] = {1, (size_t)c.isize, (size_t)c.isize * c.jsize};
  const int k0 = c.three_d ? c.gw : 0, k1 = c.three_d ? c.ksize - c.gw : 1;
  for (int v = v0; v < v1; ++v)
    for (int k = k0; k < k1; k++)
      for (int j = c.gw; j < c.jsize - c.gw; j++)
        for (int i = c.gw; i < c.isize - c.gw; i++) {
          const size_t o = c.idx(i, j, k);
          for (int D = 0; D < ND; ++D) U[o + v * N] += (F[D][o + v * N] - F[D][o + st[D] + v * N]);
        }
}

}  // namespace

void dissipative_stage(const Ctx& c, double* U, double dt, double totalTime) {
  const rgpu_params& p = c.p;
  const double nu = p.nu, eta = p.mhdEnabled ? p.eta : 0.0;
  if (!(nu > 0 || eta > 0)) return;
  const int ND = c.three_d ? 3 : 2;
  const size_t N = c.ncell;
  const int gw = c.gw, isize = c.isize, jsize = c.jsize, ksize = c.ksize;
  const double dx = c.dx, dy = c.dy, dz = c.dz;
  // ghosts of the freshly updated state (plain fill, or the shearing-box one with the end-of-step remap time)
  make_all_boundaries(c, U, totalTime, dt);
  std::vector<double> F[3];
  for (int d = 0; d < ND; ++d) F[d].assign(N * 5, 0.0);

  if (eta > 0) {
    const size_t sj = (size_t)isize, sk = (size_t)isize * jsize;
    std::vector<double> emf(N * 3, 0.0);
    const int k0 = c.three_d ? gw : 0, k1 = c.three_d ? ksize - gw + 1 : 1;
    // J = curl B at the cell edges; emf = -eta J (times dt in the CT update)
    for (int k = k0; k < k1; k++)
      for (int j = gw; j < jsize - gw + 1; j++)
        for (int i = gw; i < isize - gw + 1; i++) {
          const size_t o = c.idx(i, j, k);
          const double dbydx = (U[o + IB * N] - U[o - 1 + IB * N]) / dx;
          const double dbxdy = (U[o + IA * N] - U[o - sj + IA * N]) / dy;
          if (c.three_d) {
            const double dbzdx = (U[o + IC * N] - U[o - 1 + IC * N]) / dx;
            const double dbzdy = (U[o + IC * N] - U[o - sj + IC * N]) / dy;
            const double dbxdz = (U[o + IA * N] - U[o - sk + IA * N]) / dz;
            const double dbydz = (U[o + IB * N] - U[o - sk + IB * N]) / dz;
            emf[o + I_EMFX * N] = -eta * (dbzdy - dbydz);
            emf[o + I_EMFY * N] = -eta * (dbxdz - dbzdx);
          }
          emf[o + I_EMFZ * N] = -eta * (dbydx - dbxdy);
        }
    // constrained transport with the resistive emf (compute_ct_update_{2d,3d})
    const double dtdx = dt / dx, dtdy = dt / dy, dtdz = dt / dz;
    for (int k = k0; k < k1; k++)
      for (int j = gw; j < jsize - gw + 1; j++)
        for (int i = gw; i < isize - gw + 1; i++) {
          const size_t o = c.idx(i, j, k);
          if (!c.three_d) {
            U[o + IA * N] += (emf[o + sj + I_EMFZ * N] - emf[o + I_EMFZ * N]) * dtdy;
            U[o + IB * N] -= (emf[o + 1 + I_EMFZ * N] - emf[o + I_EMFZ * N]) * dtdx;
          } else {
            if (k < ksize - gw) {
              U[o + IA * N] += (emf[o + sj + I_EMFZ * N] - emf[o + I_EMFZ * N]) * dtdy;
              U[o + IB * N] -= (emf[o + 1 + I_EMFZ * N] - emf[o + I_EMFZ * N]) * dtdx;
            }
            U[o + IA * N] -= (emf[o + sk + I_EMFY * N] - emf[o + I_EMFY * N]) * dtdz;
            U[o + IB * N] += (emf[o + sk + I_EMFX * N] - emf[o + I_EMFX * N]) * dtdz;
            U[o + IC * N] += (emf[o + 1 + I_EMFY * N] - emf[o + I_EMFY * N]) * dtdx;
            U[o + IC * N] -= (emf[o + sj + I_EMFX * N] - emf[o + I_EMFX * N]) * dtdy;
          }
        }
    if (p.cIso <= 0) {
      // energy flux -eta (J x B) . n at the faces, from the field AFTER the resistive CT update
      const double* A = U + IA * N; const double* B = U + IB * N; const double* C = U + IC * N;
      for (int k = k0; k < k1; k++)
        for (int j = gw; j < jsize - gw + 1; j++)
          for (int i = gw; i < isize - gw + 1; i++) {
            const size_t o = c.idx(i, j, k);
            double bx, by, bz, jx, jy, jz, jxp1, jyp1, jzp1;
            if (!c.three_d) {
              by = (B[o] + B[o - 1] + B[o + sj] + B[o - 1 + sj]) / 4;
              bz = (C[o] + C[o - 1]) / 2;
              jy = -(C[o] - C[o - 1]) / dx;
              jz = (B[o] - B[o - 1]) / dx - (A[o] - A[o - sj]) / dy;
              jzp1 = (B[o + sj] - B[o - 1 + sj]) / dx - (A[o + sj] - A[o]) / dy;
              jz = (jz + jzp1) / 2;
              F[0][o + IP * N] = -eta * (jy * bz - jz * by) * dt / dx;
              bx = (A[o] + A[o - sj] + A[o + 1] + A[o + 1 - sj]) / 4;
              bz = (C[o] + C[o - sj]) / 2;
              jx = (C[o] - C[o - sj]) / dy;
              jz = (B[o] - B[o - 1]) / dx - (A[o] - A[o - sj]) / dy;
              jzp1 = (B[o + 1] - B[o]) / dx - (A[o + 1] - A[o + 1 - sj]) / dy;
              jz = (jz + jzp1) / 2;
              F[1][o + IP * N] = -eta * (jz * bx - jx * bz) * dt / dy;
            } else {
              by = (B[o] + B[o - 1] + B[o + sj] + B[o - 1 + sj]) / 4;
              bz = (C[o] + C[o - 1] + C[o + sk] + C[o - 1 + sk]) / 4;
              jy = (A[o] - A[o - sk]) / dz - (C[o] - C[o - 1]) / dx;
              jyp1 = (A[o + sk] - A[o]) / dz - (C[o + sk] - C[o - 1 + sk]) / dx;
              jy = (jy + jyp1) / 2;
              jz = (B[o] - B[o - 1]) / dx - (A[o] - A[o - sj]) / dy;
              jzp1 = (B[o + sj] - B[o - 1 + sj]) / dx - (A[o + sj] - A[o]) / dy;
              jz = (jz + jzp1) / 2;
              F[0][o + IP * N] = -eta * (jy * bz - jz * by) * dt / dx;
              bx = (A[o] + A[o - sj] + A[o + 1] + A[o + 1 - sj]) / 4;
              bz = (C[o] + C[o - sj] + C[o + sk] + C[o - sj + sk]) / 4;
              jx = (C[o] - C[o - sj]) / dy - (B[o] - B[o - sk]) / dz;
              jxp1 = (C[o + sk] - C[o - sj + sk]) / dy - (B[o + sk] - B[o]) / dz;
              jx = (jx + jxp1) / 2;
              jz = (B[o] - B[o - 1]) / dx - (A[o] - A[o - sj]) / dy;
              jzp1 = (B[o + 1] - B[o]) / dx - (A[o + 1] - A[o + 1 - sj]) / dy;
              jz = (jz + jzp1) / 2;
              F[1][o + IP * N] = -eta * (jz * bx - jx * bz) * dt / dy;
              bx = (A[o] + A[o - sk] + A[o + 1] + A[o + 1 - sk]) / 4;
              by = (B[o] + B[o - sk] + B[o + sj] + B[o + sj - sk]) / 4;
              jx = (C[o] - C[o - sj]) / dy - (B[o] - B[o - sk]) / dz;
              jxp1 = (C[o + sj] - C[o]) / dy - (B[o + sj] - B[o + sj - sk]) / dz;
              jx = (jx + jxp1) / 2;
              jy = (A[o] - A[o - sk]) / dz - (C[o] - C[o - 1]) / dx;
              jyp1 = (A[o + 1] - A[o + 1 - sk]) / dz - (C[o + 1] - C[o]) / dx;
              jy = (jy + jyp1) / 2;
              F[2][o + IP * N] = -eta * (jx * by - jy * bx) * dt / dz;
            }
          }
      flux_update(c, U, F, IP, IP + 1);   // compute_hydro_update_energy
    }
  }

  if (nu > 0) {
    viscosity_flux(c, U, F, dt);
    flux_update(c, U, F, 0, ND + 2);    // compute_hydro_update: rho, E and the ND momenta
  }
}

// compute_random_forcing_normalization + add_random_forcing, the reference's sequential loops (k, j, i); of the nine
// sums of the reference only the two that enter the normalisation are kept (the others are debug output)
void random_forcing(const Ctx& c, double* U, double dt) {
  const rgpu_params& p = c.p;
  if (!p.randomForcingEnabled || !c.three_d || !c.Frc) return;
  const int gw = c.gw;
  const size_t N = c.ncell;
  const double* F = c.Frc;
  double r0 = 0.0, r1 = 0.0;
  const long long nbCells = (long long)p.nx * p.ny * p.nz;
  for (int k = gw; k < c.ksize - gw; k++)
    for (int j = gw; j < c.jsize - gw; j++)
      for (int i = gw; i < c.isize - gw; i++) {
        const size_t o = c.idx(i, j, k);
        const double rho = U[o + ID * N];
        const double u = U[o + IU * N] / rho, v = U[o + IV * N] / rho, w = U[o + IW * N] / rho;
        const double uu = F[o], vv = F[o + N], ww = F[o + 2 * N];
        r0 += rho * (u * uu + v * vv + w * ww);
        r1 += rho * uu * uu;
        r1 += rho * vv * vv;
        r1 += rho * ww * ww;
      }
  double norm;
  if (p.randomForcingEdot == 0) norm = 0;
  else norm = (std::sqrt(r0 * r0 + r1 * dt * p.randomForcingEdot * 2 * nbCells) - r0) / r1;
  for (int k = gw; k < c.ksize - gw; k++)
    for (int j = gw; j < c.jsize - gw; j++)
      for (int i = gw; i < c.isize - gw; i++) {
        const size_t o = c.idx(i, j, k);
        const double rho = U[o + ID * N];
        U[o + IP * N] += U[o + IU * N] / rho * F[o] * norm + 0.5 * ((F[o] * norm) * (F[o] * norm));
        U[o + IP * N] += U[o + IV * N] / rho * F[o + N] * norm + 0.5 * ((F[o + N] * norm) * (F[o + N] * norm));
        U[o + IP * N] += U[o + IW * N] / rho * F[o + 2 * N] * norm + 0.5 * ((F[o + 2 * N] * norm) * (F[o + 2 * N] * norm));
        U[o + IU * N] += rho * F[o] * norm;
        U[o + IV * N] += rho * F[o + N] * norm;
        U[o + IW * N] += rho * F[o + 2 * N] * norm;
      }
}

}  // namespace orc
