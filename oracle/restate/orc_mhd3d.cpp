// orc_mhd3d.cpp -- ORACLE (test infrastructure).  3D MHD unsplit step.
//   plain    : godunov_unsplit_cpu (MHDRunGodunov.cpp:1447-1503) + godunov_unsplit_cpu_v3
//              (mhd_godunov_unsplit_cpu_v3.cpp:28-631), i.e. implementationVersion 3 / 4
//   rotating : godunov_unsplit_rotating_cpu, 3D branch (MHDRunGodunov.cpp:2031-2083, 2436-3440) incl. the
//              shearing-box flux / emf remap (:3203-3300) and the ghost fill at the END of the step (:3428-3438)
//   trace_unsplit_mhd_3d_simpler      trace_mhd.h:1854-2248
//   convertToPrimitives (3D)          MHDRunGodunov.cpp:519-560
#include "orc_pointwise.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <thread>
#include <algorithm>
#include <string>
#include <sched.h>

namespace orc {

namespace {

void trace_mhd_3d(const rgpu_params& g, const double q[8], double dq[3][8], const double bfNb[6], const double dbf[12],
                  const double E[3][2][2], double dtdx, double dtdy, double dtdz, double xPos, double qm[3][8],
                  double qp[3][8], double qEdge[4][3][8]) {
  double* qRT_X = qEdge[0][0]; double* qRB_X = qEdge[1][0]; double* qLT_X = qEdge[2][0]; double* qLB_X = qEdge[3][0];
  double* qRT_Y = qEdge[0][1]; double* qRB_Y = qEdge[1][1]; double* qLT_Y = qEdge[2][1]; double* qLB_Y = qEdge[3][1];
  double* qRT_Z = qEdge[0][2]; double* qRB_Z = qEdge[1][2]; double* qLT_Z = qEdge[2][2]; double* qLB_Z = qEdge[3][2];
  const double gamma = g.gamma0, smallR = g.smallr, smallp = g.smallp, Omega0 = g.Omega0, dx = g.dx;
  const double ELL = E[IX][0][0], ELR = E[IX][0][1], ERL = E[IX][1][0], ERR = E[IX][1][1];
  const double FLL = E[IY][0][0], FLR = E[IY][0][1], FRL = E[IY][1][0], FRR = E[IY][1][1];
  const double GLL = E[IZ][0][0], GLR = E[IZ][0][1], GRL = E[IZ][1][0], GRR = E[IZ][1][1];

  double r = q[ID], p = q[IP], u = q[IU], v = q[IV], w = q[IW], A = q[IA], B = q[IB], C = q[IC];
  double AL = bfNb[0], AR = bfNb[1], BL = bfNb[2], BR = bfNb[3], CL = bfNb[4], CR = bfNb[5];

  const double drx = dq[IX][ID] * 0.5, dpx = dq[IX][IP] * 0.5, dux = dq[IX][IU] * 0.5, dvx = dq[IX][IV] * 0.5,
               dwx = dq[IX][IW] * 0.5, dCx = dq[IX][IC] * 0.5, dBx = dq[IX][IB] * 0.5;
  const double dry = dq[IY][ID] * 0.5, dpy = dq[IY][IP] * 0.5, duy = dq[IY][IU] * 0.5, dvy = dq[IY][IV] * 0.5,
               dwy = dq[IY][IW] * 0.5, dCy = dq[IY][IC] * 0.5, dAy = dq[IY][IA] * 0.5;
  const double drz = dq[IZ][ID] * 0.5, dpz = dq[IZ][IP] * 0.5, duz = dq[IZ][IU] * 0.5, dvz = dq[IZ][IV] * 0.5,
               dwz = dq[IZ][IW] * 0.5, dAz = dq[IZ][IA] * 0.5, dBz = dq[IZ][IB] * 0.5;

  const double dALy = 0.5 * dbf[0], dALz = 0.5 * dbf[1], dBLx = 0.5 * dbf[2], dBLz = 0.5 * dbf[3], dCLx = 0.5 * dbf[4],
               dCLy = 0.5 * dbf[5];
  const double dARy = 0.5 * dbf[6], dARz = 0.5 * dbf[7], dBRx = 0.5 * dbf[8], dBRz = 0.5 * dbf[9], dCRx = 0.5 * dbf[10],
               dCRy = 0.5 * dbf[11];
  const double dAx = 0.5 * (AR - AL), dBy = 0.5 * (BR - BL), dCz = 0.5 * (CR - CL);

  double sr0, su0, sv0, sw0, sp0, sA0, sB0, sC0, sAL0, sAR0, sBL0, sBR0, sCL0, sCR0;
  sr0 = (-u * drx - dux * r) * dtdx + (-v * dry - dvy * r) * dtdy + (-w * drz - dwz * r) * dtdz;
  su0 = (-u * dux - (dpx + B * dBx + C * dCx) / r) * dtdx + (-v * duy + B * dAy / r) * dtdy + (-w * duz + C * dAz / r) * dtdz;
  sv0 = (-u * dvx + A * dBx / r) * dtdx + (-v * dvy - (dpy + A * dAy + C * dCy) / r) * dtdy + (-w * dvz + C * dBz / r) * dtdz;
  sw0 = (-u * dwx + A * dCx / r) * dtdx + (-v * dwy + B * dCy / r) * dtdy + (-w * dwz - (dpz + A * dAz + B * dBz) / r) * dtdz;
  sp0 = (-u * dpx - dux * gamma * p) * dtdx + (-v * dpy - dvy * gamma * p) * dtdy + (-w * dpz - dwz * gamma * p) * dtdz;
  sA0 = (u * dBy + B * duy - v * dAy - A * dvy) * dtdy + (u * dCz + C * duz - w * dAz - A * dwz) * dtdz;
  sB0 = (v * dAx + A * dvx - u * dBx - B * dux) * dtdx + (v * dCz + C * dvz - w * dBz - B * dwz) * dtdz;
  sC0 = (w * dAx + A * dwx - u * dCx - C * dux) * dtdx + (w * dBy + B * dwy - v * dCy - C * dvy) * dtdy;
  if (Omega0 > 0) {
    const double shear = -1.5 * Omega0 * xPos;
    sr0 = sr0 - shear * dry * dtdy;
    su0 = su0 - shear * duy * dtdy;
    sv0 = sv0 - shear * dvy * dtdy;
    sw0 = sw0 - shear * dwy * dtdy;
    sp0 = sp0 - shear * dpy * dtdy;
    sA0 = sA0 - shear * dAy * dtdy;
    sB0 = sB0 + (shear * dAx - 1.5 * Omega0 * A * dx) * dtdx + shear * dBz * dtdz;
    sC0 = sC0 - shear * dCy * dtdy;
  }
  sAL0 = +(GLR - GLL) * dtdy * 0.5 - (FLR - FLL) * dtdz * 0.5;
  sAR0 = +(GRR - GRL) * dtdy * 0.5 - (FRR - FRL) * dtdz * 0.5;
  sBL0 = -(GRL - GLL) * dtdx * 0.5 + (ELR - ELL) * dtdz * 0.5;
  sBR0 = -(GRR - GLR) * dtdx * 0.5 + (ERR - ERL) * dtdz * 0.5;
  sCL0 = +(FRL - FLL) * dtdx * 0.5 - (ERL - ELL) * dtdy * 0.5;
  sCR0 = +(FRR - FLR) * dtdx * 0.5 - (ERR - ELR) * dtdy * 0.5;

  r = r + sr0; u = u + su0; v = v + sv0; w = w + sw0; p = p + sp0; A = A + sA0; B = B + sB0; C = C + sC0;
  AL = AL + sAL0; AR = AR + sAR0; BL = BL + sBL0; BR = BR + sBR0; CL = CL + sCL0; CR = CR + sCR0;

  // 3D floors: p >= smallp WITHOUT the density factor (trace_mhd.h:2042 ...)
#define ORC_FLOOR3(s) s[ID] = fmax(smallR, s[ID]); s[IP] = fmax(smallp, s[IP])
  qp[0][ID] = r - drx; qp[0][IU] = u - dux; qp[0][IV] = v - dvx; qp[0][IW] = w - dwx; qp[0][IP] = p - dpx;
  qp[0][IA] = AL; qp[0][IB] = B - dBx; qp[0][IC] = C - dCx; ORC_FLOOR3(qp[0]);
  qm[0][ID] = r + drx; qm[0][IU] = u + dux; qm[0][IV] = v + dvx; qm[0][IW] = w + dwx; qm[0][IP] = p + dpx;
  qm[0][IA] = AR; qm[0][IB] = B + dBx; qm[0][IC] = C + dCx; ORC_FLOOR3(qm[0]);
  qp[1][ID] = r - dry; qp[1][IU] = u - duy; qp[1][IV] = v - dvy; qp[1][IW] = w - dwy; qp[1][IP] = p - dpy;
  qp[1][IA] = A - dAy; qp[1][IB] = BL; qp[1][IC] = C - dCy; ORC_FLOOR3(qp[1]);
  qm[1][ID] = r + dry; qm[1][IU] = u + duy; qm[1][IV] = v + dvy; qm[1][IW] = w + dwy; qm[1][IP] = p + dpy;
  qm[1][IA] = A + dAy; qm[1][IB] = BR; qm[1][IC] = C + dCy; ORC_FLOOR3(qm[1]);
  qp[2][ID] = r - drz; qp[2][IU] = u - duz; qp[2][IV] = v - dvz; qp[2][IW] = w - dwz; qp[2][IP] = p - dpz;
  qp[2][IA] = A - dAz; qp[2][IB] = B - dBz; qp[2][IC] = CL; ORC_FLOOR3(qp[2]);
  qm[2][ID] = r + drz; qm[2][IU] = u + duz; qm[2][IV] = v + dvz; qm[2][IW] = w + dwz; qm[2][IP] = p + dpz;
  qm[2][IA] = A + dAz; qm[2][IB] = B + dBz; qm[2][IC] = CR; ORC_FLOOR3(qm[2]);

  // X edges
  qRT_X[ID] = r + (+dry + drz); qRT_X[IU] = u + (+duy + duz); qRT_X[IV] = v + (+dvy + dvz); qRT_X[IW] = w + (+dwy + dwz);
  qRT_X[IP] = p + (+dpy + dpz); qRT_X[IA] = A + (+dAy + dAz); qRT_X[IB] = BR + (+dBRz); qRT_X[IC] = CR + (+dCRy); ORC_FLOOR3(qRT_X);
  qRB_X[ID] = r + (+dry - drz); qRB_X[IU] = u + (+duy - duz); qRB_X[IV] = v + (+dvy - dvz); qRB_X[IW] = w + (+dwy - dwz);
  qRB_X[IP] = p + (+dpy - dpz); qRB_X[IA] = A + (+dAy - dAz); qRB_X[IB] = BR + (-dBRz); qRB_X[IC] = CL + (+dCLy); ORC_FLOOR3(qRB_X);
  qLT_X[ID] = r + (-dry + drz); qLT_X[IU] = u + (-duy + duz); qLT_X[IV] = v + (-dvy + dvz); qLT_X[IW] = w + (-dwy + dwz);
  qLT_X[IP] = p + (-dpy + dpz); qLT_X[IA] = A + (-dAy + dAz); qLT_X[IB] = BL + (+dBLz); qLT_X[IC] = CR + (-dCRy); ORC_FLOOR3(qLT_X);
  qLB_X[ID] = r + (-dry - drz); qLB_X[IU] = u + (-duy - duz); qLB_X[IV] = v + (-dvy - dvz); qLB_X[IW] = w + (-dwy - dwz);
  qLB_X[IP] = p + (-dpy - dpz); qLB_X[IA] = A + (-dAy - dAz); qLB_X[IB] = BL + (-dBLz); qLB_X[IC] = CL + (-dCLy); ORC_FLOOR3(qLB_X);
  // Y edges
  qRT_Y[ID] = r + (+drx + drz); qRT_Y[IU] = u + (+dux + duz); qRT_Y[IV] = v + (+dvx + dvz); qRT_Y[IW] = w + (+dwx + dwz);
  qRT_Y[IP] = p + (+dpx + dpz); qRT_Y[IA] = AR + (+dARz); qRT_Y[IB] = B + (+dBx + dBz); qRT_Y[IC] = CR + (+dCRx); ORC_FLOOR3(qRT_Y);
  qRB_Y[ID] = r + (+drx - drz); qRB_Y[IU] = u + (+dux - duz); qRB_Y[IV] = v + (+dvx - dvz); qRB_Y[IW] = w + (+dwx - dwz);
  qRB_Y[IP] = p + (+dpx - dpz); qRB_Y[IA] = AR + (-dARz); qRB_Y[IB] = B + (+dBx - dBz); qRB_Y[IC] = CL + (+dCLx); ORC_FLOOR3(qRB_Y);
  qLT_Y[ID] = r + (-drx + drz); qLT_Y[IU] = u + (-dux + duz); qLT_Y[IV] = v + (-dvx + dvz); qLT_Y[IW] = w + (-dwx + dwz);
  qLT_Y[IP] = p + (-dpx + dpz); qLT_Y[IA] = AL + (+dALz); qLT_Y[IB] = B + (-dBx + dBz); qLT_Y[IC] = CR + (-dCRx); ORC_FLOOR3(qLT_Y);
  qLB_Y[ID] = r + (-drx - drz); qLB_Y[IU] = u + (-dux - duz); qLB_Y[IV] = v + (-dvx - dvz); qLB_Y[IW] = w + (-dwx - dwz);
  qLB_Y[IP] = p + (-dpx - dpz); qLB_Y[IA] = AL + (-dALz); qLB_Y[IB] = B + (-dBx - dBz); qLB_Y[IC] = CL + (-dCLx); ORC_FLOOR3(qLB_Y);
  // Z edges
  qRT_Z[ID] = r + (+drx + dry); qRT_Z[IU] = u + (+dux + duy); qRT_Z[IV] = v + (+dvx + dvy); qRT_Z[IW] = w + (+dwx + dwy);
  qRT_Z[IP] = p + (+dpx + dpy); qRT_Z[IA] = AR + (+dARy); qRT_Z[IB] = BR + (+dBRx); qRT_Z[IC] = C + (+dCx + dCy); ORC_FLOOR3(qRT_Z);
  qRB_Z[ID] = r + (+drx - dry); qRB_Z[IU] = u + (+dux - duy); qRB_Z[IV] = v + (+dvx - dvy); qRB_Z[IW] = w + (+dwx - dwy);
  qRB_Z[IP] = p + (+dpx - dpy); qRB_Z[IA] = AR + (-dARy); qRB_Z[IB] = BL + (+dBLx); qRB_Z[IC] = C + (+dCx - dCy); ORC_FLOOR3(qRB_Z);
  qLT_Z[ID] = r + (-drx + dry); qLT_Z[IU] = u + (-dux + duy); qLT_Z[IV] = v + (-dvx + dvy); qLT_Z[IW] = w + (-dwx + dwy);
  qLT_Z[IP] = p + (-dpx + dpy); qLT_Z[IA] = AL + (+dALy); qLT_Z[IB] = BR + (-dBRx); qLT_Z[IC] = C + (-dCx + dCy); ORC_FLOOR3(qLT_Z);
  qLB_Z[ID] = r + (-drx - dry); qLB_Z[IU] = u + (-dux - duy); qLB_Z[IV] = v + (-dvx - dvy); qLB_Z[IW] = w + (-dwx - dwy);
  qLB_Z[IP] = p + (-dpx - dpy); qLB_Z[IA] = AL + (-dALy); qLB_Z[IB] = BL + (-dBLx); qLB_Z[IC] = C + (-dCx - dCy); ORC_FLOOR3(qLB_Z);
#undef ORC_FLOOR3
}


// ------------------------------------------------------------------------------------------------------------------------
// Threaded variant of the same step for the all-cores CPU baseline: every loop nest is cut into
// contiguous z-slabs, one per thread.  The reference's own OpenMP build parallelises these loops too
// (mhd_godunov_unsplit_cpu_v3.cpp:32-35, 368-371) but its flux loop SCATTERS into the neighbours' cells and races; here the
// fluxes are stored and each cell GATHERS its six contributions in the order the sequential scatter loop delivers them
// (Coriolis, +Fx, +Fy, +Fz of its own iteration, then -Fx(i+1), -Fy(j+1), -Fz(k+1)): deterministic, and bit-identical to
// mhd_step_3d for any thread count (tests/test_oracle_golden.py::test_threaded_step_equals_sequential).  Scope: the
// configurations of the bench (no gravity, no dissipative stage, no forcing) -- orc_run_mt refuses the others.
using orc::slabs;   // orc_common.h: [k0, k1) cut into contiguous pieces, one (optionally pinned) std::thread each

}  // namespace

void mhd_step_3d(const Ctx& c, double* Uold_d, double* Unew_d, double dt, double totalTime) {
  const rgpu_params& p = c.p;
  const bool rot = p.Omega0 > 0;
  const bool shearbox = p.shearingBoxEnabled != 0;
  const int gw = c.gw, isize = c.isize, jsize = c.jsize, ksize = c.ksize, nx = c.nx, ny = c.ny;
  const double dx = c.dx, dy = c.dy;
  const double dtdx = dt / c.dx, dtdy = dt / c.dy, dtdz = dt / c.dz;
  const size_t N = c.ncell;
  const double Omega0 = p.Omega0;

  double lambda = 0, ratio = 1, alpha1 = 1, alpha2 = 0;
  if (rot) {
    lambda = Omega0 * dt;
    lambda = 0.25 * lambda * lambda;
    ratio = (1.0 - lambda) / (1.0 + lambda);
    alpha1 = 1.0 / (1.0 + lambda);
    alpha2 = Omega0 * dt / (1.0 + lambda);
  } else {
    make_all_boundaries(c, Uold_d, 0.0, 0.0);  // plain path: ghosts of the INPUT at step start
  }
  std::memcpy(Unew_d, Uold_d, sizeof(double) * N * 8);

  Field U, Unew, Q, elec, dA, dB, dC, qm[3], qp[3], qE[4][3], emf;
  U.wrap(c, Uold_d, 8); Unew.wrap(c, Unew_d, 8);
  Q.alloc(c, 8); elec.alloc(c, 3); dA.alloc(c, 3); dB.alloc(c, 3); dC.alloc(c, 3); emf.alloc(c, 3);
  for (int d = 0; d < 3; ++d) { qm[d].alloc(c, 8); qp[d].alloc(c, 8); }
  for (int e = 0; e < 4; ++e) for (int d = 0; d < 3; ++d) qE[e][d].alloc(c, 8);

  // primitive variables
  for (int k = 0; k < ksize - 1; k++)
    for (int j = 0; j < jsize - 1; j++)
      for (int i = 0; i < isize - 1; i++) {
        double u[8], q[8], cs;
        for (int v = 0; v < 8; ++v) u[v] = U(i, j, k, v);
        const double bnb[3] = {U(i + 1, j, k, IA), U(i, j + 1, k, IB), U(i, j, k + 1, IC)};
        mhd_constoprim(p, u, bnb, q, cs, dt);
        for (int v = 0; v < 8; ++v) Q(i, j, k, v) = q[v];
      }

  // edge-centred electric field
  for (int k = 1; k < ksize - 1; k++)
    for (int j = 1; j < jsize - 1; j++)
      for (int i = 1; i < isize - 1; i++) {
        double u, v, w, A, B, C;
        const double xPos = p.xMin + dx / 2 + (i - gw) * dx;
        v = 0.25 * (Q(i, j - 1, k - 1, IV) + Q(i, j - 1, k, IV) + Q(i, j, k - 1, IV) + Q(i, j, k, IV));
        w = 0.25 * (Q(i, j - 1, k - 1, IW) + Q(i, j - 1, k, IW) + Q(i, j, k - 1, IW) + Q(i, j, k, IW));
        B = 0.5 * (U(i, j, k - 1, IB) + U(i, j, k, IB));
        C = 0.5 * (U(i, j - 1, k, IC) + U(i, j, k, IC));
        elec(i, j, k, IX) = v * C - w * B;
        if (rot) { const double shear = -1.5 * Omega0 * xPos; elec(i, j, k, IX) += shear * C; }
        u = 0.25 * (Q(i - 1, j, k - 1, IU) + Q(i - 1, j, k, IU) + Q(i, j, k - 1, IU) + Q(i, j, k, IU));
        w = 0.25 * (Q(i - 1, j, k - 1, IW) + Q(i - 1, j, k, IW) + Q(i, j, k - 1, IW) + Q(i, j, k, IW));
        A = 0.5 * (U(i, j, k - 1, IA) + U(i, j, k, IA));
        C = 0.5 * (U(i - 1, j, k, IC) + U(i, j, k, IC));
        elec(i, j, k, IY) = w * A - u * C;
        u = 0.25 * (Q(i - 1, j - 1, k, IU) + Q(i - 1, j, k, IU) + Q(i, j - 1, k, IU) + Q(i, j, k, IU));
        v = 0.25 * (Q(i - 1, j - 1, k, IV) + Q(i - 1, j, k, IV) + Q(i, j - 1, k, IV) + Q(i, j, k, IV));
        A = 0.5 * (U(i, j - 1, k, IA) + U(i, j, k, IA));
        B = 0.5 * (U(i - 1, j, k, IB) + U(i, j, k, IB));
        elec(i, j, k, IZ) = u * B - v * A;
        if (rot) { const double shear = -1.5 * Omega0 * (xPos - dx / 2); elec(i, j, k, IZ) -= shear * A; }
      }

  // transverse slopes of the face-centred field (slope_unsplit_mhd_3d, slope_mhd.h:598-704: type capped at 2)
  {
    const double st = fmin(p.slope_type, 2.0);
    for (int k = 1; k < ksize - 1; k++)
      for (int j = 1; j < jsize - 1; j++)
        for (int i = 1; i < isize - 1; i++) {
          // component index = direction of the slope; the unused "own direction" slot stays 0
          dA(i, j, k, IX) = 0.0;
          dA(i, j, k, IY) = tvd_slope(st, U(i, j - 1, k, IA), U(i, j, k, IA), U(i, j + 1, k, IA));
          dA(i, j, k, IZ) = tvd_slope(st, U(i, j, k - 1, IA), U(i, j, k, IA), U(i, j, k + 1, IA));
          dB(i, j, k, IX) = tvd_slope(st, U(i - 1, j, k, IB), U(i, j, k, IB), U(i + 1, j, k, IB));
          dB(i, j, k, IY) = 0.0;
          dB(i, j, k, IZ) = tvd_slope(st, U(i, j, k - 1, IB), U(i, j, k, IB), U(i, j, k + 1, IB));
          dC(i, j, k, IX) = tvd_slope(st, U(i - 1, j, k, IC), U(i, j, k, IC), U(i + 1, j, k, IC));
          dC(i, j, k, IY) = tvd_slope(st, U(i, j - 1, k, IC), U(i, j, k, IC), U(i, j + 1, k, IC));
          dC(i, j, k, IZ) = 0.0;
        }
  }

  // trace
  for (int k = gw - 2; k < ksize - gw + 1; k++)
    for (int j = gw - 2; j < jsize - gw + 1; j++)
      for (int i = gw - 2; i < isize - gw + 1; i++) {
        double q[8], dq[3][8], bfNb[6], dbf[12], E[3][2][2], tqm[3][8], tqp[3][8], tqe[4][3][8];
        const double xPos = p.xMin + dx / 2 + (i - gw) * dx;
        for (int v = 0; v < 8; ++v) {
          q[v] = Q(i, j, k, v);
          if (p.slope_type == 0) { dq[IX][v] = 0.0; dq[IY][v] = 0.0; dq[IZ][v] = 0.0; }
          else if (p.slope_type == 3) {   // plain path only (mhd_godunov_unsplit_cpu_v3.cpp:196-211)
            double nb[27], d[3];
            int m = 0;
            for (int a = -1; a < 2; ++a) for (int b = -1; b < 2; ++b) for (int c = -1; c < 2; ++c) nb[m++] = Q(i + a, j + b, k + c, v);
            d[0] = 0.5 * (Q(i + 1, j, k, v) - Q(i - 1, j, k, v));
            d[1] = 0.5 * (Q(i, j + 1, k, v) - Q(i, j - 1, k, v));
            d[2] = 0.5 * (Q(i, j, k + 1, v) - Q(i, j, k - 1, v));
            const double dlim = positivity_limiter(nb, 27, q[v], d, 3);
            dq[IX][v] = dlim * d[0]; dq[IY][v] = dlim * d[1]; dq[IZ][v] = dlim * d[2];
          } else {
            dq[IX][v] = tvd_slope(p.slope_type, Q(i - 1, j, k, v), q[v], Q(i + 1, j, k, v));
            dq[IY][v] = tvd_slope(p.slope_type, Q(i, j - 1, k, v), q[v], Q(i, j + 1, k, v));
            dq[IZ][v] = tvd_slope(p.slope_type, Q(i, j, k - 1, v), q[v], Q(i, j, k + 1, v));
          }
        }
        bfNb[0] = U(i, j, k, IA); bfNb[1] = U(i + 1, j, k, IA); bfNb[2] = U(i, j, k, IB); bfNb[3] = U(i, j + 1, k, IB);
        bfNb[4] = U(i, j, k, IC); bfNb[5] = U(i, j, k + 1, IC);
        dbf[0] = dA(i, j, k, IY); dbf[1] = dA(i, j, k, IZ); dbf[2] = dB(i, j, k, IX); dbf[3] = dB(i, j, k, IZ);
        dbf[4] = dC(i, j, k, IX); dbf[5] = dC(i, j, k, IY);
        dbf[6] = dA(i + 1, j, k, IY); dbf[7] = dA(i + 1, j, k, IZ); dbf[8] = dB(i, j + 1, k, IX); dbf[9] = dB(i, j + 1, k, IZ);
        dbf[10] = dC(i, j, k + 1, IX); dbf[11] = dC(i, j, k + 1, IY);
        E[IX][0][0] = elec(i, j, k, IX); E[IX][0][1] = elec(i, j, k + 1, IX); E[IX][1][0] = elec(i, j + 1, k, IX); E[IX][1][1] = elec(i, j + 1, k + 1, IX);
        E[IY][0][0] = elec(i, j, k, IY); E[IY][0][1] = elec(i, j, k + 1, IY); E[IY][1][0] = elec(i + 1, j, k, IY); E[IY][1][1] = elec(i + 1, j, k + 1, IY);
        E[IZ][0][0] = elec(i, j, k, IZ); E[IZ][0][1] = elec(i, j + 1, k, IZ); E[IZ][1][0] = elec(i + 1, j, k, IZ); E[IZ][1][1] = elec(i + 1, j + 1, k, IZ);
        trace_mhd_3d(p, q, dq, bfNb, dbf, E, dtdx, dtdy, dtdz, xPos, tqm, tqp, tqe);
        if (p.gravityEnabled) {   // gravity predictor on all 18 traced states (..._cpu_v3.cpp:277-332, MHDRunGodunov.cpp:2684-2740)
          const double grav_x = 0.5 * dt * c.grav(i, j, k, 0), grav_y = 0.5 * dt * c.grav(i, j, k, 1), grav_z = 0.5 * dt * c.grav(i, j, k, 2);
          for (int d = 0; d < 3; ++d) {
            tqm[d][IU] += grav_x; tqm[d][IV] += grav_y; tqm[d][IW] += grav_z;
            tqp[d][IU] += grav_x; tqp[d][IV] += grav_y; tqp[d][IW] += grav_z;
            for (int e = 0; e < 4; ++e) { tqe[e][d][IU] += grav_x; tqe[e][d][IV] += grav_y; tqe[e][d][IW] += grav_z; }
          }
        }
        for (int v = 0; v < 8; ++v) {
          for (int d = 0; d < 3; ++d) { qm[d](i, j, k, v) = tqm[d][v]; qp[d](i, j, k, v) = tqp[d][v]; }
          for (int e = 0; e < 4; ++e) for (int d = 0; d < 3; ++d) qE[e][d](i, j, k, v) = tqe[e][d][v];
        }
      }

  // shearing-box border buffers: [I_DENS, I_EMF_Y] per (j,k)  (constants.h:105-114)
  std::vector<double> sf_min((size_t)jsize * ksize * 2, 0.0), sf_max((size_t)jsize * ksize * 2, 0.0);
  std::vector<double> sf_min_remap((size_t)jsize * ksize, 0.0), sf_max_remap((size_t)jsize * ksize, 0.0);
  auto SF = [&](std::vector<double>& b, int j, int k, int comp) -> double& { return b[(size_t)j + (size_t)jsize * (k + (size_t)ksize * comp)]; };

  // fluxes + scatter update + emf
  static const int perm_y[8] = {ID, IP, IV, IU, IW, IB, IA, IC};
  static const int perm_z[8] = {ID, IP, IW, IV, IU, IC, IB, IA};
  for (int k = gw; k < ksize - gw + 1; k++)
    for (int j = gw; j < jsize - gw + 1; j++)
      for (int i = gw; i < isize - gw + 1; i++) {
        double ql[8], qr[8], flux_x[8], flux_y[8], flux_z[8];
        const double xPos = p.xMin + dx / 2 + (i - gw) * dx;
        for (int v = 0; v < 8; ++v) { flux_x[v] = 0.0; flux_y[v] = 0.0; flux_z[v] = 0.0; }
        for (int v = 0; v < 8; ++v) { ql[v] = qm[0](i - 1, j, k, v); qr[v] = qp[0](i, j, k, v); }
        mhd_riemann(p, ql, qr, flux_x);
        for (int v = 0; v < 8; ++v) { ql[v] = qm[1](i, j - 1, k, perm_y[v]); qr[v] = qp[1](i, j, k, perm_y[v]); }
        mhd_riemann(p, ql, qr, flux_y);
        if (rot) {  // shear advection of the y flux, with the states as MODIFIED by the Riemann solver
          const double shear_y = -1.5 * Omega0 * xPos;
          double eMag, eKin, eTot;
          const double bn_mean = 0.5 * (ql[IA] + qr[IA]);
          const double gamma = p.gamma0;
          const double* s = (shear_y > 0) ? ql : qr;
          eMag = 0.5 * (s[IA] * s[IA] + s[IB] * s[IB] + s[IC] * s[IC]);
          eKin = 0.5 * (s[IU] * s[IU] + s[IV] * s[IV] + s[IW] * s[IW]);
          eTot = eKin + eMag + s[IP] / (gamma - 1.0);
          flux_y[ID] = flux_y[ID] + shear_y * s[ID];
          flux_y[IP] = flux_y[IP] + shear_y * (eTot + eMag - bn_mean * bn_mean);
          flux_y[IU] = flux_y[IU] + shear_y * s[ID] * s[IU];
          flux_y[IV] = flux_y[IV] + shear_y * s[ID] * s[IV];
          flux_y[IW] = flux_y[IW] + shear_y * s[ID] * s[IW];
        }
        for (int v = 0; v < 8; ++v) { ql[v] = qm[2](i, j, k - 1, perm_z[v]); qr[v] = qp[2](i, j, k, perm_z[v]); }
        mhd_riemann(p, ql, qr, flux_z);

        const bool in_i = i < isize - gw, in_j = j < jsize - gw, in_k = k < ksize - gw;
        if (rot && in_i && in_j && in_k) {  // Coriolis (MHDRunGodunov.cpp:2938-2945)
          const double dsx = 2.0 * Omega0 * dt * Unew(i, j, k, IV) / (1.0 + lambda);
          const double dsy = -0.5 * Omega0 * dt * Unew(i, j, k, IU) / (1.0 + lambda);
          Unew(i, j, k, IU) = Unew(i, j, k, IU) * ratio + dsx;
          Unew(i, j, k, IV) = Unew(i, j, k, IV) * ratio + dsy;
        }
        // x
        if (i > gw && in_j && in_k) {
          if (rot && shearbox && i == (nx + gw)) SF(sf_max, j, k, 0) = flux_x[ID] * dtdx;
          else Unew(i - 1, j, k, ID) -= flux_x[ID] * dtdx;
          Unew(i - 1, j, k, IP) -= flux_x[IP] * dtdx;
          Unew(i - 1, j, k, IU) -= (alpha1 * flux_x[IU] + alpha2 * flux_x[IV]) * dtdx;
          Unew(i - 1, j, k, IV) -= (alpha1 * flux_x[IV] - 0.25 * alpha2 * flux_x[IU]) * dtdx;
          Unew(i - 1, j, k, IW) -= flux_x[IW] * dtdx;
        }
        if (in_i && in_j && in_k) {
          if (rot && shearbox && i == gw) SF(sf_min, j, k, 0) = flux_x[ID] * dtdx;
          else Unew(i, j, k, ID) += flux_x[ID] * dtdx;
          Unew(i, j, k, IP) += flux_x[IP] * dtdx;
          Unew(i, j, k, IU) += (alpha1 * flux_x[IU] + alpha2 * flux_x[IV]) * dtdx;
          Unew(i, j, k, IV) += (alpha1 * flux_x[IV] - 0.25 * alpha2 * flux_x[IU]) * dtdx;
          Unew(i, j, k, IW) += flux_x[IW] * dtdx;
        }
        // y (IU <-> IV swapped)
        if (in_i && j > gw && in_k) {
          Unew(i, j - 1, k, ID) -= flux_y[ID] * dtdy;
          Unew(i, j - 1, k, IP) -= flux_y[IP] * dtdy;
          Unew(i, j - 1, k, IU) -= (alpha1 * flux_y[IV] + alpha2 * flux_y[IU]) * dtdy;
          Unew(i, j - 1, k, IV) -= (alpha1 * flux_y[IU] - 0.25 * alpha2 * flux_y[IV]) * dtdy;
          Unew(i, j - 1, k, IW) -= flux_y[IW] * dtdy;
        }
        if (in_i && in_j && in_k) {
          Unew(i, j, k, ID) += flux_y[ID] * dtdy;
          Unew(i, j, k, IP) += flux_y[IP] * dtdy;
          Unew(i, j, k, IU) += (alpha1 * flux_y[IV] + alpha2 * flux_y[IU]) * dtdy;
          Unew(i, j, k, IV) += (alpha1 * flux_y[IU] - 0.25 * alpha2 * flux_y[IV]) * dtdy;
          Unew(i, j, k, IW) += flux_y[IW] * dtdy;
        }
        // z (IU <-> IW swapped)
        if (in_i && in_j && k > gw) {
          Unew(i, j, k - 1, ID) -= flux_z[ID] * dtdz;
          Unew(i, j, k - 1, IP) -= flux_z[IP] * dtdz;
          Unew(i, j, k - 1, IU) -= (alpha1 * flux_z[IW] + alpha2 * flux_z[IV]) * dtdz;
          Unew(i, j, k - 1, IV) -= (alpha1 * flux_z[IV] - 0.25 * alpha2 * flux_z[IW]) * dtdz;
          Unew(i, j, k - 1, IW) -= flux_z[IU] * dtdz;
        }
        if (in_i && in_j && in_k) {
          Unew(i, j, k, ID) += flux_z[ID] * dtdz;
          Unew(i, j, k, IP) += flux_z[IP] * dtdz;
          Unew(i, j, k, IU) += (alpha1 * flux_z[IW] + alpha2 * flux_z[IV]) * dtdz;
          Unew(i, j, k, IV) += (alpha1 * flux_z[IV] - 0.25 * alpha2 * flux_z[IW]) * dtdz;
          Unew(i, j, k, IW) += flux_z[IU] * dtdz;
        }

        // electromotive forces at the three low edges of the cell
        double qe[4][8];
        for (int v = 0; v < 8; ++v) {
          qe[0][v] = qE[0][2](i - 1, j - 1, k, v); qe[1][v] = qE[1][2](i - 1, j, k, v);
          qe[2][v] = qE[2][2](i, j - 1, k, v);     qe[3][v] = qE[3][2](i, j, k, v);
        }
        const double emfZ = compute_emf<2>(p, qe, xPos);
        if (!rot || in_k) emf(i, j, k, I_EMFZ) = emfZ;
        for (int v = 0; v < 8; ++v) {  // RB and LT are swapped for emfY
          qe[0][v] = qE[0][1](i - 1, j, k - 1, v); qe[1][v] = qE[2][1](i, j, k - 1, v);
          qe[2][v] = qE[1][1](i - 1, j, k, v);     qe[3][v] = qE[3][1](i, j, k, v);
        }
        const double emfY = compute_emf<1>(p, qe, xPos);
        if (!rot || in_j) {
          emf(i, j, k, I_EMFY) = emfY;
          if (rot && shearbox) {
            if (i == gw) SF(sf_min, j, k, 1) = emfY;
            if (i == (nx + gw)) SF(sf_max, j, k, 1) = emfY;
          }
        }
        for (int v = 0; v < 8; ++v) {
          qe[0][v] = qE[0][0](i, j - 1, k - 1, v); qe[1][v] = qE[1][0](i, j - 1, k, v);
          qe[2][v] = qE[2][0](i, j, k - 1, v);     qe[3][v] = qE[3][0](i, j, k, v);
        }
        const double emfX = compute_emf<0>(p, qe, xPos);
        if (!rot || in_i) emf(i, j, k, I_EMFX) = emfX;
      }

  // gravity source term on the momenta (compute_gravity_source_term, HydroRunBase.cpp:1925-1985; called at
  // ..._cpu_v3.cpp:586-588 and, rotating path, MHDRunGodunov.cpp:3190-3192 -- before the shear remap of the density)
  if (p.gravityEnabled)
    for (int k = gw; k < ksize - gw; k++)
      for (int j = gw; j < jsize - gw; j++)
        for (int i = gw; i < isize - gw; i++) {
          const double rhoOld = U(i, j, k, ID), rhoNew = Unew(i, j, k, ID);
          Unew(i, j, k, IU) += 0.5 * dt * c.grav(i, j, k, 0) * (rhoOld + rhoNew);
          Unew(i, j, k, IV) += 0.5 * dt * c.grav(i, j, k, 1) * (rhoOld + rhoNew);
          Unew(i, j, k, IW) += 0.5 * dt * c.grav(i, j, k, 2) * (rhoOld + rhoNew);
        }

  if (rot && shearbox) {
    // flux / emf remap across the sheared x boundary (Dumses bval_shear_flux / bval_shear_emf)
    double deltay, epsi, eps;
    int jplus, jremap, jremapp1;
    deltay = 1.5 * p.Omega0 * (p.dx * p.nx) * (totalTime + dt / 2);
    deltay = fmod(deltay, (p.dy * p.ny));
    jplus = (int)(deltay / dy);
    epsi = fmod(deltay, dy);
    for (int k = 0; k < ksize; k++)
      for (int j = 0; j < jsize; j++) {
        jremap = j - jplus - 1;
        jremapp1 = jremap + 1;
        eps = 1.0 - epsi / dy;
        if (jremap < gw) jremap += ny;
        if (jremapp1 < gw) jremapp1 += ny;
        if (j >= gw && j < jsize - gw + 1 && k >= gw && k < ksize - gw + 1) {
          sf_min_remap[(size_t)j + (size_t)jsize * k] =
              SF(sf_min, j, k, 0) + (1.0 - eps) * SF(sf_max, jremap, k, 0) + eps * SF(sf_max, jremapp1, k, 0);
          sf_min_remap[(size_t)j + (size_t)jsize * k] *= 0.5;
        }
        emf(gw, j, k, I_EMFY) += (1.0 - eps) * SF(sf_max, jremap, k, 1) + eps * SF(sf_max, jremapp1, k, 1);
        emf(gw, j, k, I_EMFY) *= 0.5;

        jremap = j + jplus;
        jremapp1 = jremap + 1;
        eps = epsi / dy;
        if (jremap > ny + gw - 1) jremap -= ny;
        if (jremapp1 > ny + gw - 1) jremapp1 -= ny;
        if (j >= gw && j < jsize - gw + 1 && k >= gw && k < ksize - gw + 1) {
          sf_max_remap[(size_t)j + (size_t)jsize * k] =
              SF(sf_max, j, k, 0) + (1.0 - eps) * SF(sf_min, jremap, k, 0) + eps * SF(sf_min, jremapp1, k, 0);
          sf_max_remap[(size_t)j + (size_t)jsize * k] *= 0.5;
        }
        emf(nx + gw, j, k, I_EMFY) += (1.0 - eps) * SF(sf_min, jremap, k, 1) + eps * SF(sf_min, jremapp1, k, 1);
        emf(nx + gw, j, k, I_EMFY) *= 0.5;
      }
    for (int k = gw; k < ksize - gw + 1; k++)
      for (int j = gw; j < jsize - gw + 1; j++) {
        Unew(gw, j, k, ID) += sf_min_remap[(size_t)j + (size_t)jsize * k];
        Unew(nx + gw - 1, j, k, ID) -= sf_max_remap[(size_t)j + (size_t)jsize * k];
        Unew(gw, j, k, ID) = fmax(Unew(gw, j, k, ID), p.smallr);
        Unew(nx + gw - 1, j, k, ID) = fmax(Unew(nx + gw - 1, j, k, ID), p.smallr);
      }
  }

  // constrained transport
  for (int k = gw; k < ksize - gw + 1; k++)
    for (int j = gw; j < jsize - gw + 1; j++)
      for (int i = gw; i < isize - gw + 1; i++) {
        if (k < ksize - gw) {
          Unew(i, j, k, IA) += (emf(i, j + 1, k, I_EMFZ) - emf(i, j, k, I_EMFZ)) * dtdy;
          Unew(i, j, k, IB) -= (emf(i + 1, j, k, I_EMFZ) - emf(i, j, k, I_EMFZ)) * dtdx;
        }
        Unew(i, j, k, IA) -= (emf(i, j, k + 1, I_EMFY) - emf(i, j, k, I_EMFY)) * dtdz;
        Unew(i, j, k, IB) += (emf(i, j, k + 1, I_EMFX) - emf(i, j, k, I_EMFX)) * dtdz;
        Unew(i, j, k, IC) += (emf(i + 1, j, k, I_EMFY) - emf(i, j, k, I_EMFY)) * dtdx;
        Unew(i, j, k, IC) -= (emf(i, j + 1, k, I_EMFX) - emf(i, j, k, I_EMFX)) * dtdy;
      }

  dissipative_stage(c, Unew_d, dt, totalTime);   // nu / eta > 0 (..._cpu_v3.cpp:662-694, MHDRunGodunov.cpp:3379-3420)
  if (!rot) random_forcing(c, Unew_d, dt);       // problem "turbulence" (..._cpu_v3.cpp:696-704); the rotating step has none
  if (!rot) ou_forcing(c, Unew_d, dt);           // problem "turbulence-Ornstein-Uhlenbeck" (..._cpu_v3.cpp:706-710)
  if (rot) make_all_boundaries(c, Unew_d, totalTime, dt);  // rotating path: ghosts of the OUTPUT at step end
}

MtWork::MtWork(const Ctx& c, int nthreads) : buf(0), doubles(0) {
  const size_t N = c.ncell;
  doubles = N * (size_t)(8 + 3 * 5 + 3 * (8 + 8 + 5) + 12 * 8);
  buf = static_cast<double*>(std::malloc(doubles * sizeof(double)));
  if (!buf) throw std::bad_alloc();
  const size_t plane = (size_t)c.isize * c.jsize;
  const size_t comps = doubles / N;
  slabs(0, c.ksize, nthreads, [&](int ka, int kb) {
    for (size_t v = 0; v < comps; ++v) std::memset(buf + v * N + plane * ka, 0, sizeof(double) * plane * (kb - ka));
  });
}
MtWork::~MtWork() { std::free(buf); }

// ---- thread placement of the all-cores baseline (bench.py: cpu_baseline_all_cores) ------------------------------------------
// mode 0: threads not pinned (OS scheduler).  1: thread t of n is pinned to the CPU at fraction t / n of the list of allowed CPUs
// ordered NUMA node by NUMA node (and, inside a node, physical core by physical core with its SMT siblings adjacent) -- the
// planes at fraction f of the box are then worked on by the same NUMA node for every thread count, which is also the node that
// first touched them (MtWork, the state copies of orc_run_mt_scan).  2: the same inside NUMA node 0 only (one socket).
namespace {
struct Placement { int mode; std::vector<int> cpus; };
Placement& placement() { static Placement p = {0, std::vector<int>()}; return p; }
bool read_int(const std::string& path, long& v) {
  FILE* f = std::fopen(path.c_str(), "r");
  if (!f) return false;
  const bool ok = std::fscanf(f, "%ld", &v) == 1;
  std::fclose(f);
  return ok;
}
std::vector<int> parse_cpulist(const std::string& path) {   // "0-63,128-191"
  std::vector<int> out;
  FILE* f = std::fopen(path.c_str(), "r");
  if (!f) return out;
  char buf[4096];
  if (std::fgets(buf, sizeof(buf), f)) {
    const char* s = buf;
    while (*s) {
      char* e;
      const long a = std::strtol(s, &e, 10);
      if (e == s) break;
      long b = a;
      s = e;
      if (*s == '-') { b = std::strtol(s + 1, &e, 10); s = e; }
      for (long c = a; c <= b; ++c) out.push_back((int)c);
      if (*s == ',') ++s; else break;
    }
  }
  std::fclose(f);
  return out;
}
}  // namespace

int set_thread_placement(int mode) {
  Placement& P = placement();
  P.mode = 0;
  P.cpus.clear();
  if (mode <= 0) return 0;
  cpu_set_t allowed;
  CPU_ZERO(&allowed);
  if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return 0;
  struct Cpu { int node; long pkg, core; int id; };
  std::vector<Cpu> all;
  for (int node = 0; node < 64; ++node) {
    const std::vector<int> l = parse_cpulist("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
    for (int id : l) {
      if (id >= CPU_SETSIZE || !CPU_ISSET(id, &allowed)) continue;
      long pkg = 0, core = id;
      read_int("/sys/devices/system/cpu/cpu" + std::to_string(id) + "/topology/physical_package_id", pkg);
      read_int("/sys/devices/system/cpu/cpu" + std::to_string(id) + "/topology/core_id", core);
      const Cpu c = {node, pkg, core, id};
      all.push_back(c);
    }
  }
  if (all.empty())   // no NUMA information: the allowed CPUs in numerical order, one node
    for (int id = 0; id < CPU_SETSIZE; ++id)
      if (CPU_ISSET(id, &allowed)) { const Cpu c = {0, 0, id, id}; all.push_back(c); }
  std::sort(all.begin(), all.end(), [](const Cpu& a, const Cpu& b) {
    if (a.node != b.node) return a.node < b.node;
    if (a.pkg != b.pkg) return a.pkg < b.pkg;
    if (a.core != b.core) return a.core < b.core;
    return a.id < b.id;
  });
  const int first_node = all.empty() ? 0 : all[0].node;
  for (const Cpu& c : all)
    if (mode == 1 || c.node == first_node) P.cpus.push_back(c.id);
  P.mode = P.cpus.empty() ? 0 : mode;
  return (int)P.cpus.size();
}

void pin_slab_thread(int t, int nthreads) {
  const Placement& P = placement();
  if (P.mode == 0 || P.cpus.empty()) return;
  const size_t i = (size_t)t * P.cpus.size() / (size_t)(nthreads > 0 ? nthreads : 1);
  cpu_set_t one;
  CPU_ZERO(&one);
  CPU_SET(P.cpus[i < P.cpus.size() ? i : P.cpus.size() - 1], &one);
  (void)sched_setaffinity(0, sizeof(one), &one);   // tid 0 = the calling thread
}

void mhd_step_3d_mt(const Ctx& c, MtWork& w, double* Uold_d, double* Unew_d, double dt, double totalTime, int nthreads) {
  static const bool timing = std::getenv("ORC_MT_TIMING") != 0;
  std::chrono::steady_clock::time_point tprev = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    const std::chrono::steady_clock::time_point now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "  orc_mt %-10s %.3f s\n", what, std::chrono::duration<double>(now - tprev).count());
    tprev = now;
  };
  const rgpu_params& p = c.p;
  const bool rot = p.Omega0 > 0;
  const bool shearbox = p.shearingBoxEnabled != 0;
  const int gw = c.gw, isize = c.isize, jsize = c.jsize, ksize = c.ksize, nx = c.nx, ny = c.ny;
  const double dx = c.dx, dy = c.dy;
  const double dtdx = dt / c.dx, dtdy = dt / c.dy, dtdz = dt / c.dz;
  const size_t N = c.ncell;
  const double Omega0 = p.Omega0;
  double lambda = 0, ratio = 1, alpha1 = 1, alpha2 = 0;
  if (rot) {
    lambda = Omega0 * dt;
    lambda = 0.25 * lambda * lambda;
    ratio = (1.0 - lambda) / (1.0 + lambda);
    alpha1 = 1.0 / (1.0 + lambda);
    alpha2 = Omega0 * dt / (1.0 + lambda);
  } else {
    make_all_boundaries(c, Uold_d, 0.0, 0.0);
  }
  slabs(0, ksize, nthreads, [&](int ka, int kb) {
    const size_t plane = (size_t)isize * jsize;
    for (int v = 0; v < 8; ++v) std::memcpy(Unew_d + v * N + plane * ka, Uold_d + v * N + plane * ka, sizeof(double) * plane * (kb - ka));
  });

  // work arrays: allocated once per run and zeroed by the threads that will use each slab (first touch); a step overwrites the
  // same cells every time, the cells no loop writes keep the zeros the sequential step's fresh arrays have
  Field U, Unew, Q, elec, dA, dB, dC, qm[3], qp[3], qE[4][3], emf, Fl[3];
  U.wrap(c, Uold_d, 8); Unew.wrap(c, Unew_d, 8);
  {
    size_t o = 0;
    auto take = [&](Field& f, int nv) { f.wrap(c, w.buf + o, nv); o += N * nv; };
    take(Q, 8); take(elec, 3); take(dA, 3); take(dB, 3); take(dC, 3); take(emf, 3);
    for (int d = 0; d < 3; ++d) { take(qm[d], 8); take(qp[d], 8); take(Fl[d], 5); }
    for (int e = 0; e < 4; ++e) for (int d = 0; d < 3; ++d) take(qE[e][d], 8);
  }

  lap("copy");
  slabs(0, ksize - 1, nthreads, [&](int ka, int kb) {   // primitive variables
    for (int k = ka; k < kb; k++)
      for (int j = 0; j < jsize - 1; j++)
        for (int i = 0; i < isize - 1; i++) {
          double u[8], q[8], cs;
          for (int v = 0; v < 8; ++v) u[v] = U(i, j, k, v);
          const double bnb[3] = {U(i + 1, j, k, IA), U(i, j + 1, k, IB), U(i, j, k + 1, IC)};
          mhd_constoprim(p, u, bnb, q, cs, dt);
          for (int v = 0; v < 8; ++v) Q(i, j, k, v) = q[v];
        }
  });
  lap("prim");
  const double st_face = fmin(p.slope_type, 2.0);
  slabs(1, ksize - 1, nthreads, [&](int ka, int kb) {   // edge electric field + transverse slopes of the face field
    for (int k = ka; k < kb; k++)
      for (int j = 1; j < jsize - 1; j++)
        for (int i = 1; i < isize - 1; i++) {
          double u, v, w, A, B, C;
          const double xPos = p.xMin + dx / 2 + (i - gw) * dx;
          v = 0.25 * (Q(i, j - 1, k - 1, IV) + Q(i, j - 1, k, IV) + Q(i, j, k - 1, IV) + Q(i, j, k, IV));
          w = 0.25 * (Q(i, j - 1, k - 1, IW) + Q(i, j - 1, k, IW) + Q(i, j, k - 1, IW) + Q(i, j, k, IW));
          B = 0.5 * (U(i, j, k - 1, IB) + U(i, j, k, IB));
          C = 0.5 * (U(i, j - 1, k, IC) + U(i, j, k, IC));
          elec(i, j, k, IX) = v * C - w * B;
          if (rot) { const double shear = -1.5 * Omega0 * xPos; elec(i, j, k, IX) += shear * C; }
          u = 0.25 * (Q(i - 1, j, k - 1, IU) + Q(i - 1, j, k, IU) + Q(i, j, k - 1, IU) + Q(i, j, k, IU));
          w = 0.25 * (Q(i - 1, j, k - 1, IW) + Q(i - 1, j, k, IW) + Q(i, j, k - 1, IW) + Q(i, j, k, IW));
          A = 0.5 * (U(i, j, k - 1, IA) + U(i, j, k, IA));
          C = 0.5 * (U(i - 1, j, k, IC) + U(i, j, k, IC));
          elec(i, j, k, IY) = w * A - u * C;
          u = 0.25 * (Q(i - 1, j - 1, k, IU) + Q(i - 1, j, k, IU) + Q(i, j - 1, k, IU) + Q(i, j, k, IU));
          v = 0.25 * (Q(i - 1, j - 1, k, IV) + Q(i - 1, j, k, IV) + Q(i, j - 1, k, IV) + Q(i, j, k, IV));
          A = 0.5 * (U(i, j - 1, k, IA) + U(i, j, k, IA));
          B = 0.5 * (U(i - 1, j, k, IB) + U(i, j, k, IB));
          elec(i, j, k, IZ) = u * B - v * A;
          if (rot) { const double shear = -1.5 * Omega0 * (xPos - dx / 2); elec(i, j, k, IZ) -= shear * A; }
          dA(i, j, k, IX) = 0.0;
          dA(i, j, k, IY) = tvd_slope(st_face, U(i, j - 1, k, IA), U(i, j, k, IA), U(i, j + 1, k, IA));
          dA(i, j, k, IZ) = tvd_slope(st_face, U(i, j, k - 1, IA), U(i, j, k, IA), U(i, j, k + 1, IA));
          dB(i, j, k, IX) = tvd_slope(st_face, U(i - 1, j, k, IB), U(i, j, k, IB), U(i + 1, j, k, IB));
          dB(i, j, k, IY) = 0.0;
          dB(i, j, k, IZ) = tvd_slope(st_face, U(i, j, k - 1, IB), U(i, j, k, IB), U(i, j, k + 1, IB));
          dC(i, j, k, IX) = tvd_slope(st_face, U(i - 1, j, k, IC), U(i, j, k, IC), U(i + 1, j, k, IC));
          dC(i, j, k, IY) = tvd_slope(st_face, U(i, j - 1, k, IC), U(i, j, k, IC), U(i, j + 1, k, IC));
          dC(i, j, k, IZ) = 0.0;
        }
  });
  lap("elec");
  slabs(gw - 2, ksize - gw + 1, nthreads, [&](int ka, int kb) {   // trace
    for (int k = ka; k < kb; k++)
      for (int j = gw - 2; j < jsize - gw + 1; j++)
        for (int i = gw - 2; i < isize - gw + 1; i++) {
          double q[8], dq[3][8], bfNb[6], dbf[12], E[3][2][2], tqm[3][8], tqp[3][8], tqe[4][3][8];
          const double xPos = p.xMin + dx / 2 + (i - gw) * dx;
          for (int v = 0; v < 8; ++v) {
            q[v] = Q(i, j, k, v);
            if (p.slope_type == 0) { dq[IX][v] = 0.0; dq[IY][v] = 0.0; dq[IZ][v] = 0.0; }
            else {
              dq[IX][v] = tvd_slope(p.slope_type, Q(i - 1, j, k, v), q[v], Q(i + 1, j, k, v));
              dq[IY][v] = tvd_slope(p.slope_type, Q(i, j - 1, k, v), q[v], Q(i, j + 1, k, v));
              dq[IZ][v] = tvd_slope(p.slope_type, Q(i, j, k - 1, v), q[v], Q(i, j, k + 1, v));
            }
          }
          bfNb[0] = U(i, j, k, IA); bfNb[1] = U(i + 1, j, k, IA); bfNb[2] = U(i, j, k, IB); bfNb[3] = U(i, j + 1, k, IB);
          bfNb[4] = U(i, j, k, IC); bfNb[5] = U(i, j, k + 1, IC);
          dbf[0] = dA(i, j, k, IY); dbf[1] = dA(i, j, k, IZ); dbf[2] = dB(i, j, k, IX); dbf[3] = dB(i, j, k, IZ);
          dbf[4] = dC(i, j, k, IX); dbf[5] = dC(i, j, k, IY);
          dbf[6] = dA(i + 1, j, k, IY); dbf[7] = dA(i + 1, j, k, IZ); dbf[8] = dB(i, j + 1, k, IX); dbf[9] = dB(i, j + 1, k, IZ);
          dbf[10] = dC(i, j, k + 1, IX); dbf[11] = dC(i, j, k + 1, IY);
          E[IX][0][0] = elec(i, j, k, IX); E[IX][0][1] = elec(i, j, k + 1, IX); E[IX][1][0] = elec(i, j + 1, k, IX); E[IX][1][1] = elec(i, j + 1, k + 1, IX);
          E[IY][0][0] = elec(i, j, k, IY); E[IY][0][1] = elec(i, j, k + 1, IY); E[IY][1][0] = elec(i + 1, j, k, IY); E[IY][1][1] = elec(i + 1, j, k + 1, IY);
          E[IZ][0][0] = elec(i, j, k, IZ); E[IZ][0][1] = elec(i, j + 1, k, IZ); E[IZ][1][0] = elec(i + 1, j, k, IZ); E[IZ][1][1] = elec(i + 1, j + 1, k, IZ);
          trace_mhd_3d(p, q, dq, bfNb, dbf, E, dtdx, dtdy, dtdz, xPos, tqm, tqp, tqe);
          for (int v = 0; v < 8; ++v) {
            for (int d = 0; d < 3; ++d) { qm[d](i, j, k, v) = tqm[d][v]; qp[d](i, j, k, v) = tqp[d][v]; }
            for (int e = 0; e < 4; ++e) for (int d = 0; d < 3; ++d) qE[e][d](i, j, k, v) = tqe[e][d][v];
          }
        }
  });

  lap("trace");
  std::vector<double> sf_min((size_t)jsize * ksize * 2, 0.0), sf_max((size_t)jsize * ksize * 2, 0.0);
  std::vector<double> sf_min_remap((size_t)jsize * ksize, 0.0), sf_max_remap((size_t)jsize * ksize, 0.0);
  auto SF = [&](std::vector<double>& b, int j, int k, int comp) -> double& { return b[(size_t)j + (size_t)jsize * (k + (size_t)ksize * comp)]; };
  static const int perm_y[8] = {ID, IP, IV, IU, IW, IB, IA, IC};
  static const int perm_z[8] = {ID, IP, IW, IV, IU, IC, IB, IA};
  slabs(gw, ksize - gw + 1, nthreads, [&](int ka, int kb) {   // Riemann problems: fluxes and emfs STORED (no scatter)
    for (int k = ka; k < kb; k++)
      for (int j = gw; j < jsize - gw + 1; j++)
        for (int i = gw; i < isize - gw + 1; i++) {
          double ql[8], qr[8], flux_x[8], flux_y[8], flux_z[8];
          const double xPos = p.xMin + dx / 2 + (i - gw) * dx;
          for (int v = 0; v < 8; ++v) { flux_x[v] = 0.0; flux_y[v] = 0.0; flux_z[v] = 0.0; }
          for (int v = 0; v < 8; ++v) { ql[v] = qm[0](i - 1, j, k, v); qr[v] = qp[0](i, j, k, v); }
          mhd_riemann(p, ql, qr, flux_x);
          for (int v = 0; v < 8; ++v) { ql[v] = qm[1](i, j - 1, k, perm_y[v]); qr[v] = qp[1](i, j, k, perm_y[v]); }
          mhd_riemann(p, ql, qr, flux_y);
          if (rot) {
            const double shear_y = -1.5 * Omega0 * xPos;
            double eMag, eKin, eTot;
            const double bn_mean = 0.5 * (ql[IA] + qr[IA]);
            const double gamma = p.gamma0;
            const double* s = (shear_y > 0) ? ql : qr;
            eMag = 0.5 * (s[IA] * s[IA] + s[IB] * s[IB] + s[IC] * s[IC]);
            eKin = 0.5 * (s[IU] * s[IU] + s[IV] * s[IV] + s[IW] * s[IW]);
            eTot = eKin + eMag + s[IP] / (gamma - 1.0);
            flux_y[ID] = flux_y[ID] + shear_y * s[ID];
            flux_y[IP] = flux_y[IP] + shear_y * (eTot + eMag - bn_mean * bn_mean);
            flux_y[IU] = flux_y[IU] + shear_y * s[ID] * s[IU];
            flux_y[IV] = flux_y[IV] + shear_y * s[ID] * s[IV];
            flux_y[IW] = flux_y[IW] + shear_y * s[ID] * s[IW];
          }
          for (int v = 0; v < 8; ++v) { ql[v] = qm[2](i, j, k - 1, perm_z[v]); qr[v] = qp[2](i, j, k, perm_z[v]); }
          mhd_riemann(p, ql, qr, flux_z);
          for (int v = 0; v < 5; ++v) { Fl[0](i, j, k, v) = flux_x[v]; Fl[1](i, j, k, v) = flux_y[v]; Fl[2](i, j, k, v) = flux_z[v]; }
          const bool in_i = i < isize - gw, in_j = j < jsize - gw, in_k = k < ksize - gw;
          if (rot && shearbox && in_j && in_k) {   // density flux through the two sheared x borders (taken out of the update below)
            if (i == (nx + gw)) SF(sf_max, j, k, 0) = flux_x[ID] * dtdx;
            if (i == gw) SF(sf_min, j, k, 0) = flux_x[ID] * dtdx;
          }
          double qe[4][8];
          for (int v = 0; v < 8; ++v) {
            qe[0][v] = qE[0][2](i - 1, j - 1, k, v); qe[1][v] = qE[1][2](i - 1, j, k, v);
            qe[2][v] = qE[2][2](i, j - 1, k, v);     qe[3][v] = qE[3][2](i, j, k, v);
          }
          const double emfZ = compute_emf<2>(p, qe, xPos);
          if (!rot || in_k) emf(i, j, k, I_EMFZ) = emfZ;
          for (int v = 0; v < 8; ++v) {
            qe[0][v] = qE[0][1](i - 1, j, k - 1, v); qe[1][v] = qE[2][1](i, j, k - 1, v);
            qe[2][v] = qE[1][1](i - 1, j, k, v);     qe[3][v] = qE[3][1](i, j, k, v);
          }
          const double emfY = compute_emf<1>(p, qe, xPos);
          if (!rot || in_j) {
            emf(i, j, k, I_EMFY) = emfY;
            if (rot && shearbox) {
              if (i == gw) SF(sf_min, j, k, 1) = emfY;
              if (i == (nx + gw)) SF(sf_max, j, k, 1) = emfY;
            }
          }
          for (int v = 0; v < 8; ++v) {
            qe[0][v] = qE[0][0](i, j - 1, k - 1, v); qe[1][v] = qE[1][0](i, j - 1, k, v);
            qe[2][v] = qE[2][0](i, j, k - 1, v);     qe[3][v] = qE[3][0](i, j, k, v);
          }
          const double emfX = compute_emf<0>(p, qe, xPos);
          if (!rot || in_i) emf(i, j, k, I_EMFX) = emfX;
        }
  });
  lap("riemann");
  slabs(gw, ksize - gw, nthreads, [&](int ka, int kb) {   // gather update of the interior cells, contributions in scatter order
    for (int k = ka; k < kb; k++)
      for (int j = gw; j < jsize - gw; j++)
        for (int i = gw; i < isize - gw; i++) {
          if (rot) {
            const double dsx = 2.0 * Omega0 * dt * Unew(i, j, k, IV) / (1.0 + lambda);
            const double dsy = -0.5 * Omega0 * dt * Unew(i, j, k, IU) / (1.0 + lambda);
            Unew(i, j, k, IU) = Unew(i, j, k, IU) * ratio + dsx;
            Unew(i, j, k, IV) = Unew(i, j, k, IV) * ratio + dsy;
          }
          const Field& fx = Fl[0]; const Field& fy = Fl[1]; const Field& fz = Fl[2];
          if (!(rot && shearbox && i == gw)) Unew(i, j, k, ID) += fx(i, j, k, ID) * dtdx;
          Unew(i, j, k, IP) += fx(i, j, k, IP) * dtdx;
          Unew(i, j, k, IU) += (alpha1 * fx(i, j, k, IU) + alpha2 * fx(i, j, k, IV)) * dtdx;
          Unew(i, j, k, IV) += (alpha1 * fx(i, j, k, IV) - 0.25 * alpha2 * fx(i, j, k, IU)) * dtdx;
          Unew(i, j, k, IW) += fx(i, j, k, IW) * dtdx;
          Unew(i, j, k, ID) += fy(i, j, k, ID) * dtdy;
          Unew(i, j, k, IP) += fy(i, j, k, IP) * dtdy;
          Unew(i, j, k, IU) += (alpha1 * fy(i, j, k, IV) + alpha2 * fy(i, j, k, IU)) * dtdy;
          Unew(i, j, k, IV) += (alpha1 * fy(i, j, k, IU) - 0.25 * alpha2 * fy(i, j, k, IV)) * dtdy;
          Unew(i, j, k, IW) += fy(i, j, k, IW) * dtdy;
          Unew(i, j, k, ID) += fz(i, j, k, ID) * dtdz;
          Unew(i, j, k, IP) += fz(i, j, k, IP) * dtdz;
          Unew(i, j, k, IU) += (alpha1 * fz(i, j, k, IW) + alpha2 * fz(i, j, k, IV)) * dtdz;
          Unew(i, j, k, IV) += (alpha1 * fz(i, j, k, IV) - 0.25 * alpha2 * fz(i, j, k, IW)) * dtdz;
          Unew(i, j, k, IW) += fz(i, j, k, IU) * dtdz;
          if (!(rot && shearbox && (i + 1) == (nx + gw))) Unew(i, j, k, ID) -= fx(i + 1, j, k, ID) * dtdx;
          Unew(i, j, k, IP) -= fx(i + 1, j, k, IP) * dtdx;
          Unew(i, j, k, IU) -= (alpha1 * fx(i + 1, j, k, IU) + alpha2 * fx(i + 1, j, k, IV)) * dtdx;
          Unew(i, j, k, IV) -= (alpha1 * fx(i + 1, j, k, IV) - 0.25 * alpha2 * fx(i + 1, j, k, IU)) * dtdx;
          Unew(i, j, k, IW) -= fx(i + 1, j, k, IW) * dtdx;
          Unew(i, j, k, ID) -= fy(i, j + 1, k, ID) * dtdy;
          Unew(i, j, k, IP) -= fy(i, j + 1, k, IP) * dtdy;
          Unew(i, j, k, IU) -= (alpha1 * fy(i, j + 1, k, IV) + alpha2 * fy(i, j + 1, k, IU)) * dtdy;
          Unew(i, j, k, IV) -= (alpha1 * fy(i, j + 1, k, IU) - 0.25 * alpha2 * fy(i, j + 1, k, IV)) * dtdy;
          Unew(i, j, k, IW) -= fy(i, j + 1, k, IW) * dtdy;
          Unew(i, j, k, ID) -= fz(i, j, k + 1, ID) * dtdz;
          Unew(i, j, k, IP) -= fz(i, j, k + 1, IP) * dtdz;
          Unew(i, j, k, IU) -= (alpha1 * fz(i, j, k + 1, IW) + alpha2 * fz(i, j, k + 1, IV)) * dtdz;
          Unew(i, j, k, IV) -= (alpha1 * fz(i, j, k + 1, IV) - 0.25 * alpha2 * fz(i, j, k + 1, IW)) * dtdz;
          Unew(i, j, k, IW) -= fz(i, j, k + 1, IU) * dtdz;
        }
  });

  lap("update");
  if (rot && shearbox) {   // O(N^2): sequential, as in mhd_step_3d
    double deltay, epsi, eps;
    int jplus, jremap, jremapp1;
    deltay = 1.5 * p.Omega0 * (p.dx * p.nx) * (totalTime + dt / 2);
    deltay = fmod(deltay, (p.dy * p.ny));
    jplus = (int)(deltay / dy);
    epsi = fmod(deltay, dy);
    for (int k = 0; k < ksize; k++)
      for (int j = 0; j < jsize; j++) {
        jremap = j - jplus - 1;
        jremapp1 = jremap + 1;
        eps = 1.0 - epsi / dy;
        if (jremap < gw) jremap += ny;
        if (jremapp1 < gw) jremapp1 += ny;
        if (j >= gw && j < jsize - gw + 1 && k >= gw && k < ksize - gw + 1) {
          sf_min_remap[(size_t)j + (size_t)jsize * k] =
              SF(sf_min, j, k, 0) + (1.0 - eps) * SF(sf_max, jremap, k, 0) + eps * SF(sf_max, jremapp1, k, 0);
          sf_min_remap[(size_t)j + (size_t)jsize * k] *= 0.5;
        }
        emf(gw, j, k, I_EMFY) += (1.0 - eps) * SF(sf_max, jremap, k, 1) + eps * SF(sf_max, jremapp1, k, 1);
        emf(gw, j, k, I_EMFY) *= 0.5;
        jremap = j + jplus;
        jremapp1 = jremap + 1;
        eps = epsi / dy;
        if (jremap > ny + gw - 1) jremap -= ny;
        if (jremapp1 > ny + gw - 1) jremapp1 -= ny;
        if (j >= gw && j < jsize - gw + 1 && k >= gw && k < ksize - gw + 1) {
          sf_max_remap[(size_t)j + (size_t)jsize * k] =
              SF(sf_max, j, k, 0) + (1.0 - eps) * SF(sf_min, jremap, k, 0) + eps * SF(sf_min, jremapp1, k, 0);
          sf_max_remap[(size_t)j + (size_t)jsize * k] *= 0.5;
        }
        emf(nx + gw, j, k, I_EMFY) += (1.0 - eps) * SF(sf_min, jremap, k, 1) + eps * SF(sf_min, jremapp1, k, 1);
        emf(nx + gw, j, k, I_EMFY) *= 0.5;
      }
    for (int k = gw; k < ksize - gw + 1; k++)
      for (int j = gw; j < jsize - gw + 1; j++) {
        Unew(gw, j, k, ID) += sf_min_remap[(size_t)j + (size_t)jsize * k];
        Unew(nx + gw - 1, j, k, ID) -= sf_max_remap[(size_t)j + (size_t)jsize * k];
        Unew(gw, j, k, ID) = fmax(Unew(gw, j, k, ID), p.smallr);
        Unew(nx + gw - 1, j, k, ID) = fmax(Unew(nx + gw - 1, j, k, ID), p.smallr);
      }
  }
  lap("shear");
  slabs(gw, ksize - gw + 1, nthreads, [&](int ka, int kb) {   // constrained transport
    for (int k = ka; k < kb; k++)
      for (int j = gw; j < jsize - gw + 1; j++)
        for (int i = gw; i < isize - gw + 1; i++) {
          if (k < ksize - gw) {
            Unew(i, j, k, IA) += (emf(i, j + 1, k, I_EMFZ) - emf(i, j, k, I_EMFZ)) * dtdy;
            Unew(i, j, k, IB) -= (emf(i + 1, j, k, I_EMFZ) - emf(i, j, k, I_EMFZ)) * dtdx;
          }
          Unew(i, j, k, IA) -= (emf(i, j, k + 1, I_EMFY) - emf(i, j, k, I_EMFY)) * dtdz;
          Unew(i, j, k, IB) += (emf(i, j, k + 1, I_EMFX) - emf(i, j, k, I_EMFX)) * dtdz;
          Unew(i, j, k, IC) += (emf(i + 1, j, k, I_EMFY) - emf(i, j, k, I_EMFY)) * dtdx;
          Unew(i, j, k, IC) -= (emf(i, j + 1, k, I_EMFX) - emf(i, j, k, I_EMFX)) * dtdy;
        }
  });
  lap("ct");
  if (rot) make_all_boundaries(c, Unew_d, totalTime, dt);
  lap("ghosts");
}

// compute_dt_mhd (3D) with the scan cut into z-slabs: max is order independent, hence the sequential value
double compute_inv_dt_mhd3d_mt(const Ctx& c, const double* U, int nthreads) {
  const rgpu_params& p = c.p;
  const int gw = c.gw;
  const size_t N = c.ncell;
  const double deltaX = p.xMax - p.xMin;
  const size_t sj = c.isize, sk = (size_t)c.isize * c.jsize;
  std::vector<double> part((size_t)(nthreads > 0 ? nthreads : 1) + c.ksize, p.smallc / fmin(c.dx, c.dy));
  slabs(gw, c.ksize - gw, nthreads, [&](int ka, int kb) {
    double invDt = p.smallc / fmin(c.dx, c.dy);
    for (int k = ka; k < kb; k++)
      for (int j = gw; j < c.jsize - gw; j++)
        for (int i = gw; i < c.isize - gw; i++) {
          const size_t o = c.idx(i, j, k);
          double u[8];
          for (int v = 0; v < 8; ++v) u[v] = U[o + v * N];
          const double bnb[3] = {U[o + 1 + IA * N], U[o + sj + IB * N], U[o + sk + IC * N]};
          double q[8], cs;
          mhd_constoprim(p, u, bnb, q, cs, 0.0);
          double s[3];
          find_speed_info<3>(p, q, s);
          double vy = s[IY];
          if (p.Omega0 > 0) vy += 1.5 * p.Omega0 * deltaX / 2;
          invDt = fmax(invDt, s[IX] / c.dx + vy / c.dy + s[IZ] / c.dz);
        }
    part[ka] = invDt;   // one slot per slab start plane: disjoint
  });
  double invDt = p.smallc / fmin(c.dx, c.dy);
  for (double v : part) invDt = fmax(invDt, v);
  return invDt;
}


}  // namespace orc
