// orc_mhd2d.cpp -- ORACLE (test infrastructure).  2D MHD unsplit step, "implementationVersion 1".
//   godunov_unsplit_cpu               MHDRunGodunov.cpp:1447-1503 (ghost fill of the input, copy, primitives)
//   convertToPrimitives (2D)          MHDRunGodunov.cpp:479-517   (i<isize-1, j<jsize-1; Bz_cell = Bz/2)
//   godunov_unsplit_cpu_v1 (2D)       mhd_godunov_unsplit_cpu_v1.cpp:35-241
//   trace_unsplit_mhd_2d              trace_mhd.h:38-339
//   godunov_unsplit_rotating_cpu (2D) MHDRunGodunov.cpp:2031-2434, 3420-3436 (Omega0 > 0: Coriolis + alpha mixing in
//                                     the update, shear terms on the Bz fluxes and emfZ, ghost fill at the END of the step)
#include "orc_pointwise.h"

namespace orc {

namespace {

void trace_mhd_2d(const rgpu_params& g, double qNb[3][3][8], double bfNb[4][4][3], double dtdx, double dtdy, double xPos,
                  double qm[2][8], double qp[2][8], double qEdge[4][8]) {
  enum { CENTER = 1 };
  double* qRT = qEdge[0]; double* qRB = qEdge[1]; double* qLT = qEdge[2]; double* qLB = qEdge[3];
  const double smallR = g.smallr, smallp = g.smallp, gamma = g.gamma0, Omega0 = g.Omega0;
  const double* q = qNb[CENTER][CENTER];

  double Ez[2][2];
  for (int di = 0; di < 2; di++)
    for (int dj = 0; dj < 2; dj++) {
      const int cx = CENTER + di, cy = CENTER + dj;
      const double u = 0.25 * (qNb[cx - 1][cy - 1][IU] + qNb[cx - 1][cy][IU] + qNb[cx][cy - 1][IU] + qNb[cx][cy][IU]);
      const double v = 0.25 * (qNb[cx - 1][cy - 1][IV] + qNb[cx - 1][cy][IV] + qNb[cx][cy - 1][IV] + qNb[cx][cy][IV]);
      const double A = 0.5 * (bfNb[cx][cy - 1][IX] + bfNb[cx][cy][IX]);
      const double B = 0.5 * (bfNb[cx - 1][cy][IY] + bfNb[cx][cy][IY]);
      Ez[di][dj] = u * B - v * A;
    }
  const double ELL = Ez[0][0], ELR = Ez[0][1], ERL = Ez[1][0], ERR = Ez[1][1];

  double r = q[ID], p = q[IP], u = q[IU], v = q[IV], w = q[IW], A = q[IA], B = q[IB], C = q[IC];
  double AL = bfNb[CENTER][CENTER][IX], AR = bfNb[CENTER + 1][CENTER][IX];
  double BL = bfNb[CENTER][CENTER][IY], BR = bfNb[CENTER][CENTER + 1][IY];

  // hydro slopes of all 8 primitive variables (slope_unsplit_hydro_2d, slope_mhd.h:77-172; types 0,1,2,3)
  double dq[2][8];
  for (int n = 0; n < 8; ++n) {
    if (g.slope_type == 0) { dq[IX][n] = 0.0; dq[IY][n] = 0.0; }
    else if (g.slope_type == 3) {
      double nb[9], d[2];
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) nb[3 * a + b] = qNb[a][b][n];
      d[0] = 0.5 * (qNb[CENTER + 1][CENTER][n] - qNb[CENTER - 1][CENTER][n]);
      d[1] = 0.5 * (qNb[CENTER][CENTER + 1][n] - qNb[CENTER][CENTER - 1][n]);
      const double dlim = positivity_limiter(nb, 9, q[n], d, 2);
      dq[IX][n] = dlim * d[0]; dq[IY][n] = dlim * d[1];
    } else {
      dq[IX][n] = tvd_slope(g.slope_type, qNb[CENTER - 1][CENTER][n], q[n], qNb[CENTER + 1][CENTER][n]);
      dq[IY][n] = tvd_slope(g.slope_type, qNb[CENTER][CENTER - 1][n], q[n], qNb[CENTER][CENTER + 1][n]);
    }
  }
  const double drx = dq[IX][ID] * 0.5, dpx = dq[IX][IP] * 0.5, dux = dq[IX][IU] * 0.5, dvx = dq[IX][IV] * 0.5,
               dwx = dq[IX][IW] * 0.5, dCx = dq[IX][IC] * 0.5, dBx = dq[IX][IB] * 0.5;
  const double dry = dq[IY][ID] * 0.5, dpy = dq[IY][IP] * 0.5, duy = dq[IY][IU] * 0.5, dvy = dq[IY][IV] * 0.5,
               dwy = dq[IY][IW] * 0.5, dCy = dq[IY][IC] * 0.5, dAy = dq[IY][IA] * 0.5;

  // transverse slopes of the face-centred field (slope_unsplit_mhd_2d, slope_mhd.h:524-574: slope_type NOT capped)
  const double st = g.slope_type;
  const double dALy = 0.5 * tvd_slope(st, bfNb[CENTER][CENTER - 1][IX], bfNb[CENTER][CENTER][IX], bfNb[CENTER][CENTER + 1][IX]);
  const double dBLx = 0.5 * tvd_slope(st, bfNb[CENTER - 1][CENTER][IY], bfNb[CENTER][CENTER][IY], bfNb[CENTER + 1][CENTER][IY]);
  const double dARy = 0.5 * tvd_slope(st, bfNb[CENTER + 1][CENTER - 1][IX], bfNb[CENTER + 1][CENTER][IX], bfNb[CENTER + 1][CENTER + 1][IX]);
  const double dBRx = 0.5 * tvd_slope(st, bfNb[CENTER - 1][CENTER + 1][IY], bfNb[CENTER][CENTER + 1][IY], bfNb[CENTER + 1][CENTER + 1][IY]);

  const double dAx = 0.5 * (AR - AL);
  const double dBy = 0.5 * (BR - BL);

  double sr0, su0, sv0, sw0, sp0, sA0, sB0, sC0, sAL0, sAR0, sBL0, sBR0;
  sr0 = (-u * drx - dux * r) * dtdx + (-v * dry - dvy * r) * dtdy;
  su0 = (-u * dux - dpx / r - B * dBx / r - C * dCx / r) * dtdx + (-v * duy + B * dAy / r) * dtdy;
  sv0 = (-u * dvx + A * dBx / r) * dtdx + (-v * dvy - dpy / r - A * dAy / r - C * dCy / r) * dtdy;
  sw0 = (-u * dwx + A * dCx / r) * dtdx + (-v * dwy + B * dCy / r) * dtdy;
  sp0 = (-u * dpx - dux * gamma * p) * dtdx + (-v * dpy - dvy * gamma * p) * dtdy;
  sA0 = (u * dBy + B * duy - v * dAy - A * dvy) * dtdy;
  sB0 = (-u * dBx - B * dux + v * dAx + A * dvx) * dtdx;
  sC0 = (w * dAx + A * dwx - u * dCx - C * dux) * dtdx + (-v * dCy - C * dvy + w * dBy + B * dwy) * dtdy;
  if (Omega0 > 0.0) {
    const double shear = -1.5 * Omega0 * xPos;
    sC0 += (shear * dAx - 1.5 * Omega0 * A) * dtdx;
    sC0 += shear * dBy * dtdy;
  }
  sAL0 = +(ELR - ELL) * 0.5 * dtdy;
  sAR0 = +(ERR - ERL) * 0.5 * dtdy;
  sBL0 = -(ERL - ELL) * 0.5 * dtdx;
  sBR0 = -(ERR - ELR) * 0.5 * dtdx;

  r = r + sr0; u = u + su0; v = v + sv0; w = w + sw0; p = p + sp0; A = A + sA0; B = B + sB0; C = C + sC0;
  AL = AL + sAL0; AR = AR + sAR0; BL = BL + sBL0; BR = BR + sBR0;

#define ORC_FLOOR(s) s[ID] = fmax(smallR, s[ID]); s[IP] = fmax(smallp * s[ID], s[IP])
  qp[0][ID] = r - drx; qp[0][IU] = u - dux; qp[0][IV] = v - dvx; qp[0][IW] = w - dwx; qp[0][IP] = p - dpx;
  qp[0][IA] = AL; qp[0][IB] = B - dBx; qp[0][IC] = C - dCx; ORC_FLOOR(qp[0]);
  qm[0][ID] = r + drx; qm[0][IU] = u + dux; qm[0][IV] = v + dvx; qm[0][IW] = w + dwx; qm[0][IP] = p + dpx;
  qm[0][IA] = AR; qm[0][IB] = B + dBx; qm[0][IC] = C + dCx; ORC_FLOOR(qm[0]);
  qp[1][ID] = r - dry; qp[1][IU] = u - duy; qp[1][IV] = v - dvy; qp[1][IW] = w - dwy; qp[1][IP] = p - dpy;
  qp[1][IA] = A - dAy; qp[1][IB] = BL; qp[1][IC] = C - dCy; ORC_FLOOR(qp[1]);
  qm[1][ID] = r + dry; qm[1][IU] = u + duy; qm[1][IV] = v + dvy; qm[1][IW] = w + dwy; qm[1][IP] = p + dpy;
  qm[1][IA] = A + dAy; qm[1][IB] = BR; qm[1][IC] = C + dCy; ORC_FLOOR(qm[1]);

  qRT[ID] = r + (+drx + dry); qRT[IU] = u + (+dux + duy); qRT[IV] = v + (+dvx + dvy); qRT[IW] = w + (+dwx + dwy);
  qRT[IP] = p + (+dpx + dpy); qRT[IA] = AR + (+dARy); qRT[IB] = BR + (+dBRx); qRT[IC] = C + (+dCx + dCy); ORC_FLOOR(qRT);
  qRB[ID] = r + (+drx - dry); qRB[IU] = u + (+dux - duy); qRB[IV] = v + (+dvx - dvy); qRB[IW] = w + (+dwx - dwy);
  qRB[IP] = p + (+dpx - dpy); qRB[IA] = AR + (-dARy); qRB[IB] = BL + (+dBLx); qRB[IC] = C + (+dCx - dCy); ORC_FLOOR(qRB);
  qLB[ID] = r + (-drx - dry); qLB[IU] = u + (-dux - duy); qLB[IV] = v + (-dvx - dvy); qLB[IW] = w + (-dwx - dwy);
  qLB[IP] = p + (-dpx - dpy); qLB[IA] = AL + (-dALy); qLB[IB] = BL + (-dBLx); qLB[IC] = C + (-dCx - dCy); ORC_FLOOR(qLB);
  qLT[ID] = r + (-drx + dry); qLT[IU] = u + (-dux + duy); qLT[IV] = v + (-dvx + dvy); qLT[IW] = w + (-dwx + dwy);
  qLT[IP] = p + (-dpx + dpy); qLT[IA] = AL + (+dALy); qLT[IB] = BR + (-dBRx); qLT[IC] = C + (-dCx + dCy); ORC_FLOOR(qLT);
#undef ORC_FLOOR
}

}  // namespace

void mhd_step_2d(const Ctx& c, double* Uold_d, double* Unew_d, double dt) {
  // Implementation versions 0 and 1 of the 2D step compute the same numbers (version 0 recomputes what version 1
  // stores); only version 0 has the static-gravity terms.
  const rgpu_params& p = c.p;
  const int gw = c.gw, isize = c.isize, jsize = c.jsize;
  const double dtdx = dt / c.dx, dtdy = dt / c.dy;
  const size_t N = c.ncell;
  const bool rot = p.Omega0 > 0;   // the 2D branch of the rotating step has no gravity terms
  const bool grav = p.gravityEnabled && p.implementationVersion == 0 && !rot;
  double lambda = 0, ratio = 1, alpha1 = 1, alpha2 = 0;
  if (rot) {   // MHDRunGodunov.cpp:2039-2053
    lambda = p.Omega0 * dt;
    lambda = 0.25 * lambda * lambda;
    ratio = (1.0 - lambda) / (1.0 + lambda);
    alpha1 = 1.0 / (1.0 + lambda);
    alpha2 = p.Omega0 * dt / (1.0 + lambda);
  }

  if (!rot) make_all_boundaries(c, Uold_d, 0.0, 0.0);
  std::memcpy(Unew_d, Uold_d, sizeof(double) * N * 8);

  Field U, Unew, Q, qm_x, qm_y, qp_x, qp_y, eRT, eRB, eLT, eLB, emf;
  U.wrap(c, Uold_d, 8); Unew.wrap(c, Unew_d, 8);
  Q.alloc(c, 8); qm_x.alloc(c, 8); qm_y.alloc(c, 8); qp_x.alloc(c, 8); qp_y.alloc(c, 8);
  eRT.alloc(c, 8); eRB.alloc(c, 8); eLT.alloc(c, 8); eLB.alloc(c, 8); emf.alloc(c, 1);

  // primitives; Q(isize-1,*) and Q(*,jsize-1) are never written (stay 0 here; unconsumed)
  for (int j = 0; j < jsize - 1; j++)
    for (int i = 0; i < isize - 1; i++) {
      double u[8], q[8], cs;
      for (int v = 0; v < 8; ++v) u[v] = U(i, j, v);
      const double bnb[3] = {U(i + 1, j, IA), U(i, j + 1, IB), 0.0};
      mhd_constoprim(p, u, bnb, q, cs, dt);
      for (int v = 0; v < 8; ++v) Q(i, j, v) = q[v];
    }

  // trace
  for (int j = gw - 2; j < jsize - gw + 2; j++)
    for (int i = gw - 2; i < isize - gw + 2; i++) {
      double qNb[3][3][8], bfNb[4][4][3], qm[2][8], qp[2][8], qEdge[4][8];
      const double xPos = p.xMin + c.dx / 2 + (i - gw) * c.dx;
      for (int di = 0; di < 3; di++)
        for (int dj = 0; dj < 3; dj++)
          for (int v = 0; v < 8; ++v) qNb[di][dj][v] = Q(i + di - 1, j + dj - 1, v);
      for (int di = 0; di < 4; di++)
        for (int dj = 0; dj < 4; dj++) {
          // the 4x4 stencil reaches i+2 / j+2: at the last traced cells that is one past the array end in the
          // reference (an over-wide loop whose result is never consumed); clamp instead of reading out of bounds
          const int ii = (i + di - 1 < isize) ? i + di - 1 : isize - 1;
          const int jj = (j + dj - 1 < jsize) ? j + dj - 1 : jsize - 1;
          bfNb[di][dj][IX] = U(ii, jj, IA); bfNb[di][dj][IY] = U(ii, jj, IB); bfNb[di][dj][IZ] = U(ii, jj, IC);
        }
      trace_mhd_2d(p, qNb, bfNb, dtdx, dtdy, xPos, qm, qp, qEdge);
      if (grav) {   // implementation version 0 only (mhd_godunov_unsplit_cpu_v0.cpp:495-525): predictor on all 8 states
        const double gx = 0.5 * dt * c.grav(i, j, 0, 0), gy = 0.5 * dt * c.grav(i, j, 0, 1);
        // the y states are already in the swapped (y-normal) frame when the reference adds the predictor
        // (swap at :177-179, :388-390, predictor at :500-512): g_x lands on v and g_y on u there
        qm[0][IU] += gx; qm[0][IV] += gy; qp[0][IU] += gx; qp[0][IV] += gy;
        qm[1][IV] += gx; qm[1][IU] += gy; qp[1][IV] += gx; qp[1][IU] += gy;
        for (int e = 0; e < 4; ++e) { qEdge[e][IU] += gx; qEdge[e][IV] += gy; }
      }
      for (int v = 0; v < 8; ++v) {
        qm_x(i, j, v) = qm[0][v]; qp_x(i, j, v) = qp[0][v]; qm_y(i, j, v) = qm[1][v]; qp_y(i, j, v) = qp[1][v];
        eRT(i, j, v) = qEdge[0][v]; eRB(i, j, v) = qEdge[1][v]; eLT(i, j, v) = qEdge[2][v]; eLB(i, j, v) = qEdge[3][v];
      }
    }

  // fluxes, scatter update (NO guards in 2D: ghost cells get written too), emfZ
  for (int j = gw; j < jsize - gw + 1; j++)
    for (int i = gw; i < isize - gw + 1; i++) {
      double ql[8], qr[8], flux_x[8], flux_y[8];
      for (int v = 0; v < 8; ++v) { flux_x[v] = 0.0; flux_y[v] = 0.0; }
      for (int v = 0; v < 8; ++v) { ql[v] = qm_x(i - 1, j, v); qr[v] = qp_x(i, j, v); }
      mhd_riemann(p, ql, qr, flux_x);
      const double xPos = p.xMin + c.dx / 2 + (i - gw) * c.dx;
      if (rot) {   // shear correction with the mean normal field left in the states by the solver (:2190-2197)
        const double shear_x = -1.5 * p.Omega0 * (xPos + xPos - c.dx);
        flux_x[IC] += shear_x * (ql[IA] + qr[IA]) / 2;
      }
      static const int perm_y[8] = {ID, IP, IV, IU, IW, IB, IA, IC};
      for (int v = 0; v < 8; ++v) { ql[v] = qm_y(i, j - 1, perm_y[v]); qr[v] = qp_y(i, j, perm_y[v]); }
      mhd_riemann(p, ql, qr, flux_y);
      if (rot) {   // :2224-2231
        const double shear_y = -1.5 * p.Omega0 * xPos;
        flux_y[IC] += shear_y * (ql[IA] + qr[IA]) / 2;
      }

      if (rot) {   // :2236-2290 (no shearing box in 2D)
        if (i < isize - gw && j < jsize - gw) {
          const double dsx = 2.0 * p.Omega0 * dt * Unew(i, j, IV) / (1.0 + lambda);
          const double dsy = -0.5 * p.Omega0 * dt * Unew(i, j, IU) / (1.0 + lambda);
          Unew(i, j, IU) = Unew(i, j, IU) * ratio + dsx;
          Unew(i, j, IV) = Unew(i, j, IV) * ratio + dsy;
        }
        Unew(i - 1, j, ID) -= flux_x[ID] * dtdx;
        Unew(i - 1, j, IP) -= flux_x[IP] * dtdx;
        Unew(i - 1, j, IU) -= (alpha1 * flux_x[IU] + alpha2 * flux_x[IV]) * dtdx;
        Unew(i - 1, j, IV) -= (alpha1 * flux_x[IV] - 0.25 * alpha2 * flux_x[IU]) * dtdx;
        Unew(i - 1, j, IW) -= flux_x[IW] * dtdx;
        Unew(i - 1, j, IC) -= flux_x[IC] * dtdx;
        Unew(i, j, ID) += flux_x[ID] * dtdx;
        Unew(i, j, IP) += flux_x[IP] * dtdx;
        Unew(i, j, IU) += (alpha1 * flux_x[IU] + alpha2 * flux_x[IV]) * dtdx;
        Unew(i, j, IV) += (alpha1 * flux_x[IV] - 0.25 * alpha2 * flux_x[IU]) * dtdx;
        Unew(i, j, IW) += flux_x[IW] * dtdx;
        Unew(i, j, IC) += flux_x[IC] * dtdx;
        Unew(i, j - 1, ID) -= flux_y[ID] * dtdy;
        Unew(i, j - 1, IP) -= flux_y[IP] * dtdy;
        Unew(i, j - 1, IU) -= (alpha1 * flux_y[IV] + alpha2 * flux_y[IU]) * dtdy;
        Unew(i, j - 1, IV) -= (alpha1 * flux_y[IU] - 0.25 * alpha2 * flux_y[IV]) * dtdy;
        Unew(i, j - 1, IW) -= flux_y[IW] * dtdy;
        Unew(i, j - 1, IC) -= flux_y[IC] * dtdy;
        Unew(i, j, ID) += flux_y[ID] * dtdy;
        Unew(i, j, IP) += flux_y[IP] * dtdy;
        Unew(i, j, IU) += (alpha1 * flux_y[IV] + alpha2 * flux_y[IU]) * dtdy;
        Unew(i, j, IV) += (alpha1 * flux_y[IU] - 0.25 * alpha2 * flux_y[IV]) * dtdy;
        Unew(i, j, IW) += flux_y[IW] * dtdy;
        Unew(i, j, IC) += flux_y[IC] * dtdy;
      } else {
      Unew(i - 1, j, ID) -= flux_x[ID] * dtdx; Unew(i - 1, j, IP) -= flux_x[IP] * dtdx; Unew(i - 1, j, IU) -= flux_x[IU] * dtdx;
      Unew(i - 1, j, IV) -= flux_x[IV] * dtdx; Unew(i - 1, j, IW) -= flux_x[IW] * dtdx; Unew(i - 1, j, IC) -= flux_x[IC] * dtdx;
      Unew(i, j, ID) += flux_x[ID] * dtdx; Unew(i, j, IP) += flux_x[IP] * dtdx; Unew(i, j, IU) += flux_x[IU] * dtdx;
      Unew(i, j, IV) += flux_x[IV] * dtdx; Unew(i, j, IW) += flux_x[IW] * dtdx; Unew(i, j, IC) += flux_x[IC] * dtdx;
      Unew(i, j - 1, ID) -= flux_y[ID] * dtdy; Unew(i, j - 1, IP) -= flux_y[IP] * dtdy; Unew(i, j - 1, IU) -= flux_y[IV] * dtdy;
      Unew(i, j - 1, IV) -= flux_y[IU] * dtdy; Unew(i, j - 1, IW) -= flux_y[IW] * dtdy; Unew(i, j - 1, IC) -= flux_y[IC] * dtdy;
      Unew(i, j, ID) += flux_y[ID] * dtdy; Unew(i, j, IP) += flux_y[IP] * dtdy; Unew(i, j, IU) += flux_y[IV] * dtdy;
      Unew(i, j, IV) += flux_y[IU] * dtdy; Unew(i, j, IW) += flux_y[IW] * dtdy; Unew(i, j, IC) += flux_y[IC] * dtdy;
      }

      double qe[4][8];
      for (int v = 0; v < 8; ++v) {
        qe[0][v] = eRT(i - 1, j - 1, v); qe[1][v] = eRB(i - 1, j, v); qe[2][v] = eLT(i, j - 1, v); qe[3][v] = eLB(i, j, v);
      }
      emf(i, j, 0) = compute_emf<2>(p, qe, xPos);
    }

  // gravity source term on the momenta (version 0: mhd_godunov_unsplit_cpu_v0.cpp:616-618, HydroRunBase.cpp:1925-1950)
  if (grav)
    for (int j = gw; j < jsize - gw; j++)
      for (int i = gw; i < isize - gw; i++) {
        const double rhoOld = U(i, j, ID), rhoNew = Unew(i, j, ID);
        Unew(i, j, IU) += 0.5 * dt * c.grav(i, j, 0, 0) * (rhoOld + rhoNew);
        Unew(i, j, IV) += 0.5 * dt * c.grav(i, j, 0, 1) * (rhoOld + rhoNew);
      }

  // constrained transport
  for (int j = gw; j < jsize - gw + 1; j++)
    for (int i = gw; i < isize - gw + 1; i++) {
      Unew(i, j, IA) += (emf(i, j + 1, 0) - emf(i, j, 0)) * dtdy;
      Unew(i, j, IB) -= (emf(i + 1, j, 0) - emf(i, j, 0)) * dtdx;
    }
  dissipative_stage(c, Unew_d, dt, 0.0);   // nu / eta > 0 (mhd_godunov_unsplit_cpu_v1.cpp:244-272)
  if (rot) make_all_boundaries(c, Unew_d, 0.0, 0.0);   // rotating path: ghosts of the OUTPUT (MHDRunGodunov.cpp:3424-3434)
}

}  // namespace orc
