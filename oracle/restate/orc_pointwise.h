// orc_pointwise.h -- ORACLE (test infrastructure).  Cell-wise numerics of the reference restated on plain doubles.
// Each function names the reference routine whose arithmetic (operand order included) it reproduces.
#pragma once
#include "orc_common.h"

namespace orc {

// ---------------------------------------------------------------------------------------------------------
// equation of state / primitive variables
// ---------------------------------------------------------------------------------------------------------

// constoprim_2D / constoprim_3D + eos (constoprim.h:29-111); NV = 4 or 5
template <int NV>
inline void hydro_constoprim(const rgpu_params& g, const double u[NV], double q[NV], double& c) {
  q[ID] = fmax(u[ID], g.smallr);
  q[IU] = u[IU] / q[ID];
  q[IV] = u[IV] / q[ID];
  if (NV == 5) q[IW] = u[IW] / q[ID];
  double eken;
  if (NV == 5)
    eken = 0.5 * (q[IU] * q[IU] + q[IV] * q[IV] + q[IW] * q[IW]);
  else
    eken = 0.5 * (q[IU] * q[IU] + q[IV] * q[IV]);
  if (g.cIso > 0) {
    q[IP] = q[ID] * g.cIso * g.cIso;
    c = g.cIso;
  } else {
    const double eint = u[IP] / q[ID] - eken;
    q[IP] = fmax((g.gamma0 - 1.0) * q[ID] * eint, q[ID] * g.smallp);
    c = sqrt(g.gamma0 * q[IP] / q[ID]);
  }
}

// constoprim_mhd (constoprim.h:137-199); bnb = left-face field of the +1 neighbours (0 for z in 2D)
inline void mhd_constoprim(const rgpu_params& g, const double u[8], const double bnb[3], double q[8], double& c, double dt) {
  q[ID] = fmax(u[ID], g.smallr);
  q[IU] = u[IU] / q[ID];
  q[IV] = u[IV] / q[ID];
  q[IW] = u[IW] / q[ID];
  q[IA] = 0.5 * (u[IA] + bnb[0]);
  q[IB] = 0.5 * (u[IB] + bnb[1]);
  q[IC] = 0.5 * (u[IC] + bnb[2]);
  const double eken = 0.5 * (q[IU] * q[IU] + q[IV] * q[IV] + q[IW] * q[IW]);
  const double emag = 0.5 * (q[IA] * q[IA] + q[IB] * q[IB] + q[IC] * q[IC]);
  if (g.cIso > 0) {
    q[IP] = q[ID] * g.cIso * g.cIso;
    c = g.cIso;
  } else {
    const double eint = (u[IP] - emag) / q[ID] - eken;
    q[IP] = fmax((g.gamma0 - 1.0) * q[ID] * eint, q[ID] * g.smallp);
    c = sqrt(g.gamma0 * q[IP] / q[ID]);
  }
  if (g.Omega0 > 0) {  // Coriolis predictor, both increments from the un-updated velocities
    const double dvx = 2.0 * g.Omega0 * q[IV];
    const double dvy = -0.5 * g.Omega0 * q[IU];
    q[IU] += dvx * dt * 0.5;
    q[IV] += dvy * dt * 0.5;
  }
}

// ---------------------------------------------------------------------------------------------------------
// slopes
// ---------------------------------------------------------------------------------------------------------

// the limiter shared by slope_unsplit_hydro_2d (slope.h:97-147), slope_unsplit_3d type 2 (slope.h:386-425),
// slope_unsplit_hydro_{2d,3d} of slope_mhd.h (:101-129, :459-500) and slope_unsplit_mhd_{2d,3d} (:549-571, :636-700)
inline double tvd_slope(double st, double qminus, double q, double qplus) {
  const double dlft = st * (q - qminus);
  const double drgt = st * (qplus - q);
  const double dcen = 0.5 * (qplus - qminus);
  const double dsgn = (dcen >= 0.0) ? 1.0 : -1.0;
  const double slop = fmin(fabs(dlft), fabs(drgt));
  double dlim = slop;
  if ((dlft * drgt) <= 0.0) dlim = 0.0;
  return dsgn * fmin(dlim, fabs(dcen));
}

// slope_unsplit_3d, slope_type == 1 branch (slope.h:351-384)
inline double minmod_slope(double qminus, double q, double qplus) {
  const double dlft = q - qminus;
  const double drgt = qplus - q;
  if ((dlft * drgt) <= 0.0) return 0.0;
  if (dlft > 0) return fmin(dlft, drgt);
  return fmax(dlft, drgt);
}

// ---------------------------------------------------------------------------------------------------------
// hydro trace (trace.h:332-414, 544-661): NDIM = 2,3 ; NV = 4,5.  dq holds FULL slopes on entry.
// ---------------------------------------------------------------------------------------------------------
template <int NDIM, int NV>
inline void hydro_trace(const rgpu_params& g, const double q[NV], double dq[NDIM][NV], double dtdx, double dtdy,
                        double dtdz, double qm[NDIM][NV], double qp[NDIM][NV]) {
  const double smallR = g.smallr, smallp = g.smallp, gamma = g.gamma0;
  double r = q[ID], p = q[IP], u = q[IU], v = q[IV], w = (NV == 5) ? q[IW] : 0.0;
  for (int d = 0; d < NDIM; ++d)
    for (int n = 0; n < NV; ++n) dq[d][n] *= 0.5;
  const double drx = dq[IX][ID], dpx = dq[IX][IP], dux = dq[IX][IU], dvx = dq[IX][IV];
  const double dry = dq[IY][ID], dpy = dq[IY][IP], duy = dq[IY][IU], dvy = dq[IY][IV];
  double sr0, su0, sv0, sw0 = 0, sp0;
  if (NDIM == 2) {
    sr0 = (-u * drx - dux * r) * dtdx + (-v * dry - dvy * r) * dtdy;
    su0 = (-u * dux - dpx / r) * dtdx + (-v * duy) * dtdy;
    sv0 = (-u * dvx) * dtdx + (-v * dvy - dpy / r) * dtdy;
    sp0 = (-u * dpx - dux * gamma * p) * dtdx + (-v * dpy - dvy * gamma * p) * dtdy;
  } else {
    const double dwx = dq[IX][IW], dwy = dq[IY][IW];
    const double drz = dq[NDIM - 1][ID], dpz = dq[NDIM - 1][IP], duz = dq[NDIM - 1][IU], dvz = dq[NDIM - 1][IV],
                 dwz = dq[NDIM - 1][IW];
    sr0 = (-u * drx - dux * r) * dtdx + (-v * dry - dvy * r) * dtdy + (-w * drz - dwz * r) * dtdz;
    su0 = (-u * dux - dpx / r) * dtdx + (-v * duy) * dtdy + (-w * duz) * dtdz;
    sv0 = (-u * dvx) * dtdx + (-v * dvy - dpy / r) * dtdy + (-w * dvz) * dtdz;
    sw0 = (-u * dwx) * dtdx + (-v * dwy) * dtdy + (-w * dwz - dpz / r) * dtdz;
    sp0 = (-u * dpx - dux * gamma * p) * dtdx + (-v * dpy - dvy * gamma * p) * dtdy + (-w * dpz - dwz * gamma * p) * dtdz;
  }
  r = r + sr0;
  u = u + su0;
  v = v + sv0;
  w = w + sw0;
  p = p + sp0;
  for (int d = 0; d < NDIM; ++d) {
    qp[d][ID] = r - dq[d][ID];
    qp[d][IU] = u - dq[d][IU];
    qp[d][IV] = v - dq[d][IV];
    if (NV == 5) qp[d][IW] = w - dq[d][IW];
    qp[d][IP] = p - dq[d][IP];
    qp[d][ID] = fmax(smallR, qp[d][ID]);
    qp[d][IP] = fmax(smallp * qp[d][ID], qp[d][IP]);
    qm[d][ID] = r + dq[d][ID];
    qm[d][IU] = u + dq[d][IU];
    qm[d][IV] = v + dq[d][IV];
    if (NV == 5) qm[d][IW] = w + dq[d][IW];
    qm[d][IP] = p + dq[d][IP];
    qm[d][ID] = fmax(smallR, qm[d][ID]);
    qm[d][IP] = fmax(smallp * qm[d][ID], qm[d][IP]);
  }
}

// ---------------------------------------------------------------------------------------------------------
// hydro Riemann solvers (riemann.h), NV = 4,5 ; states are in the face-normal frame
// ---------------------------------------------------------------------------------------------------------

// saturate_cpu (gpu_macros.cpp:25-30): the argument goes through FLOAT
inline float saturate_float(float a) {
  if (a != a) return 0.0f;
  return a >= 1.0f ? 1.0f : a <= 0.0f ? 0.0f : a;
}

// cmpflx (cmpflx.h:21-48)
template <int NV>
inline void hydro_cmpflx(const rgpu_params& g, const double qg[NV], double flux[NV]) {
  flux[ID] = qg[ID] * qg[IU];
  flux[IU] = flux[ID] * qg[IU] + qg[IP];
  flux[IV] = flux[ID] * qg[IV];
  if (NV == 5) flux[IW] = flux[ID] * qg[IW];
  const double entho = 1.0 / (g.gamma0 - 1.0);
  double ekin;
  if (NV == 5)
    ekin = 0.5 * qg[ID] * (qg[IU] * qg[IU] + qg[IV] * qg[IV] + qg[IW] * qg[IW]);
  else
    ekin = 0.5 * qg[ID] * (qg[IU] * qg[IU] + qg[IV] * qg[IV]);
  const double etot = qg[IP] * entho + ekin;
  flux[IP] = qg[IU] * (etot + qg[IP]);
}

// riemann_approx (riemann.h:29-159)
template <int NV>
inline void riemann_approx(const rgpu_params& g, const double ql[NV], const double qr[NV], double flux[NV]) {
  const double rl = fmax(ql[ID], g.smallr), ul = ql[IU], pl = fmax(ql[IP], rl * g.smallp);
  const double rr = fmax(qr[ID], g.smallr), ur = qr[IU], pr = fmax(qr[IP], rr * g.smallp);
  const double cl = g.gamma0 * pl * rl, cr = g.gamma0 * pr * rr;
  double wl = sqrt(cl), wr = sqrt(cr);
  double pstar = fmax(((wr * pl + wl * pr) + wl * wr * (ul - ur)) / (wl + wr), 0.0);
  double pold = pstar, conv = 1.0;
  for (int iter = 0; iter < g.niter_riemann && conv > 1e-6; ++iter) {
    const double wwl = sqrt(cl * (1.0 + g.gamma6 * (pold - pl) / pl));
    const double wwr = sqrt(cr * (1.0 + g.gamma6 * (pold - pr) / pr));
    const double qql = 2.0 * wwl * wwl * wwl / (wwl * wwl + cl);
    const double qqr = 2.0 * wwr * wwr * wwr / (wwr * wwr + cr);
    const double usl = ul - (pold - pl) / wwl;
    const double usr = ur + (pold - pr) / wwr;
    const double delp = fmax(qqr * qql / (qqr + qql) * (usl - usr), -pold);
    pold = pold + delp;
    conv = fabs(delp / (pold + g.smallpp));
  }
  pstar = pold;
  wl = sqrt(cl * (1.0 + g.gamma6 * (pstar - pl) / pl));
  wr = sqrt(cr * (1.0 + g.gamma6 * (pstar - pr) / pr));
  const double ustar = 0.5 * (ul + (pl - pstar) / wl + ur - (pr - pstar) / wr);
  const double sgnm = copysign(1.0, ustar);
  double ro, uo, po, wo;
  if (sgnm > 0.0) { ro = rl; uo = ul; po = pl; wo = wl; } else { ro = rr; uo = ur; po = pr; wo = wr; }
  const double co = fmax(g.smallc, sqrt(fabs(g.gamma0 * po / ro)));
  const double rstar = fmax((double)(ro / (1.0 + ro * (po - pstar) / (wo * wo))), (double)(g.smallr));
  const double cstar = fmax(g.smallc, sqrt(fabs(g.gamma0 * pstar / rstar)));
  double spout = co - sgnm * uo;
  double spin = cstar - sgnm * ustar;
  const double ushock = wo / ro - sgnm * uo;
  if (pstar >= po) { spin = ushock; spout = ushock; }
  const double scr = fmax(spout - spin, g.smallc + fabs(spout + spin));
  double frac = 0.5 * (1.0 + (spout + spin) / scr);
  if (frac != frac) frac = 0.0; else frac = saturate_float((float)frac);
  double qg[NV];
  qg[ID] = frac * rstar + (1.0 - frac) * ro;
  qg[IU] = frac * ustar + (1.0 - frac) * uo;
  qg[IP] = frac * pstar + (1.0 - frac) * po;
  if (spout < 0.0) { qg[ID] = ro; qg[IU] = uo; qg[IP] = po; }
  if (spin > 0.0) { qg[ID] = rstar; qg[IU] = ustar; qg[IP] = pstar; }
  if (sgnm > 0.0) { qg[IV] = ql[IV]; if (NV == 5) qg[IW] = ql[IW]; }
  else { qg[IV] = qr[IV]; if (NV == 5) qg[IW] = qr[IW]; }
  hydro_cmpflx<NV>(g, qg, flux);
}

// riemann_hll (riemann.h:175-255)
template <int NV>
inline void riemann_hll(const rgpu_params& g, const double ql[NV], const double qr[NV], double flux[NV]) {
  const double entho = 1.0 / (g.gamma0 - 1.0);
  const double rl = fmax(ql[ID], g.smallr), ul = ql[IU], pl = fmax(ql[IP], rl * g.smallp);
  const double rr = fmax(qr[ID], g.smallr), ur = qr[IU], pr = fmax(qr[IP], rr * g.smallp);
  const double cl = sqrt(g.gamma0 * pl / rl), cr = sqrt(g.gamma0 * pr / rr);
  const double SL = fmin(fmin(ul, ur) - fmax(cl, cr), 0.0);
  const double SR = fmax(fmax(ul, ur) + fmax(cl, cr), 0.0);
  double uL[NV], uR[NV], fL[NV], fR[NV];
  uL[ID] = ql[ID];
  uR[ID] = qr[ID];
  uL[IP] = ql[IP] * entho + 0.5 * ql[ID] * ql[IU] * ql[IU];
  uR[IP] = qr[IP] * entho + 0.5 * qr[ID] * qr[IU] * qr[IU];
  uL[IP] += 0.5 * ql[ID] * ql[IV] * ql[IV];
  uR[IP] += 0.5 * qr[ID] * qr[IV] * qr[IV];
  if (NV == 5) {
    uL[IP] += 0.5 * ql[ID] * ql[IW] * ql[IW];
    uR[IP] += 0.5 * qr[ID] * qr[IW] * qr[IW];
  }
  uL[IU] = ql[ID] * ql[IU];
  uR[IU] = qr[ID] * qr[IU];
  uL[IV] = ql[ID] * ql[IV];
  uR[IV] = qr[ID] * qr[IV];
  if (NV == 5) { uL[IW] = ql[ID] * ql[IW]; uR[IW] = qr[ID] * qr[IW]; }
  fL[ID] = uL[IU];
  fR[ID] = uR[IU];
  fL[IP] = ql[IU] * (uL[IP] + ql[IP]);
  fR[IP] = qr[IU] * (uR[IP] + qr[IP]);
  fL[IU] = ql[IP] + uL[IU] * ql[IU];
  fR[IU] = qr[IP] + uR[IU] * qr[IU];
  fL[IV] = fL[ID] * ql[IV];
  fR[IV] = fR[ID] * qr[IV];
  if (NV == 5) { fL[IW] = fL[ID] * ql[IW]; fR[IW] = fR[ID] * qr[IW]; }
  for (int n = 0; n < NV; ++n) flux[n] = (SR * fL[n] - SL * fR[n] + SR * SL * (uR[n] - uL[n])) / (SR - SL);
}

// riemann_hllc (riemann.h:269-371)
template <int NV>
inline void riemann_hllc(const rgpu_params& g, const double ql[NV], const double qr[NV], double flux[NV]) {
  const double entho = 1.0 / (g.gamma0 - 1.0);
  const double rl = fmax(ql[ID], g.smallr), pl = fmax(ql[IP], rl * g.smallp), ul = ql[IU];
  double ecinl = 0.5 * rl * ul * ul;
  ecinl += 0.5 * rl * ql[IV] * ql[IV];
  if (NV == 5) ecinl += 0.5 * rl * ql[IW] * ql[IW];
  const double etotl = pl * entho + ecinl, ptotl = pl;
  const double rr = fmax(qr[ID], g.smallr), pr = fmax(qr[IP], rr * g.smallp), ur = qr[IU];
  double ecinr = 0.5 * rr * ur * ur;
  ecinr += 0.5 * rr * qr[IV] * qr[IV];
  if (NV == 5) ecinr += 0.5 * rr * qr[IW] * qr[IW];
  const double etotr = pr * entho + ecinr, ptotr = pr;
  const double cfastl = sqrt(fmax(g.gamma0 * pl / rl, g.smallc * g.smallc));
  const double cfastr = sqrt(fmax(g.gamma0 * pr / rr, g.smallc * g.smallc));
  const double SL = fmin(ul, ur) - fmax(cfastl, cfastr);
  const double SR = fmax(ul, ur) + fmax(cfastl, cfastr);
  const double rcl = rl * (ul - SL), rcr = rr * (SR - ur);
  const double ustar = (rcr * ur + rcl * ul + (ptotl - ptotr)) / (rcr + rcl);
  const double ptotstar = (rcr * ptotl + rcl * ptotr + rcl * rcr * (ul - ur)) / (rcr + rcl);
  const double rstarl = rl * (SL - ul) / (SL - ustar);
  const double etotstarl = ((SL - ul) * etotl - ptotl * ul + ptotstar * ustar) / (SL - ustar);
  const double rstarr = rr * (SR - ur) / (SR - ustar);
  const double etotstarr = ((SR - ur) * etotr - ptotr * ur + ptotstar * ustar) / (SR - ustar);
  double ro, uo, ptoto, etoto;
  if (SL > 0.0) { ro = rl; uo = ul; ptoto = ptotl; etoto = etotl; }
  else if (ustar > 0.0) { ro = rstarl; uo = ustar; ptoto = ptotstar; etoto = etotstarl; }
  else if (SR > 0.0) { ro = rstarr; uo = ustar; ptoto = ptotstar; etoto = etotstarr; }
  else { ro = rr; uo = ur; ptoto = ptotr; etoto = etotr; }
  flux[ID] = ro * uo;
  flux[IU] = ro * uo * uo + ptoto;
  flux[IP] = (etoto + ptoto) * uo;
  if (flux[ID] > 0.0) flux[IV] = flux[ID] * ql[IV]; else flux[IV] = flux[ID] * qr[IV];
  if (NV == 5) { if (flux[ID] > 0.0) flux[IW] = flux[ID] * ql[IW]; else flux[IW] = flux[ID] * qr[IW]; }
}

// riemann<NVAR> dispatch (riemann.h:388-401); HLLD/LLF selections leave the flux untouched
template <int NV>
inline void hydro_riemann(const rgpu_params& g, const double ql[NV], const double qr[NV], double flux[NV]) {
  if (g.riemannSolver == RGPU_RS_APPROX) riemann_approx<NV>(g, ql, qr, flux);
  else if (g.riemannSolver == RGPU_RS_HLL) riemann_hll<NV>(g, ql, qr, flux);
  else if (g.riemannSolver == RGPU_RS_HLLC) riemann_hllc<NV>(g, ql, qr, flux);
}

// ---------------------------------------------------------------------------------------------------------
// MHD utilities (mhd_utils.h)
// ---------------------------------------------------------------------------------------------------------

// find_speed_fast<dir> (mhd_utils.h:28-52)
template <int DIR>
inline double find_speed_fast(const rgpu_params& g, const double q[8]) {
  const double d = q[ID], p = q[IP], a = q[IA], b = q[IB], c = q[IC];
  const double b2 = a * a + b * b + c * c;
  const double c2 = g.gamma0 * p / d;
  const double d2 = 0.5 * (b2 / d + c2);
  if (DIR == IX) return sqrt(d2 + sqrt(d2 * d2 - c2 * a * a / d));
  if (DIR == IY) return sqrt(d2 + sqrt(d2 * d2 - c2 * b * b / d));
  return sqrt(d2 + sqrt(d2 * d2 - c2 * c * c / d));
}

// find_speed_info<NDIM> (mhd_utils.h:241-284)
template <int NDIM>
inline void find_speed_info(const rgpu_params& g, const double q[8], double out[3]) {
  const double d = q[ID], p = q[IP], a = q[IA], b = q[IB], c = q[IC];
  const double b2 = a * a + b * b + c * c;
  const double c2 = g.gamma0 * p / d;
  const double d2 = 0.5 * (b2 / d + c2);
  double cf = sqrt(d2 + sqrt(d2 * d2 - c2 * a * a / d));
  out[IX] = cf + fabs(q[IU]);
  cf = sqrt(d2 + sqrt(d2 * d2 - c2 * b * b / d));
  out[IY] = cf + fabs(q[IV]);
  if (NDIM == 3) {
    cf = sqrt(d2 + sqrt(d2 * d2 - c2 * c * c / d));
    out[IZ] = cf + fabs(q[IW]);
  }
}

// find_speed_info, x only (mhd_utils.h:295-316)
inline double find_speed_info_x(const rgpu_params& g, const double q[8]) {
  const double d = q[ID], p = q[IP], a = q[IA], b = q[IB], c = q[IC];
  const double b2 = a * a + b * b + c * c;
  const double c2 = g.gamma0 * p / d;
  const double d2 = 0.5 * (b2 / d + c2);
  const double cf = sqrt(d2 + sqrt(d2 * d2 - c2 * a * a / d));
  return cf + fabs(q[IU]);
}

// find_mhd_flux (mhd_utils.h:106-156)
inline void find_mhd_flux(const rgpu_params& g, const double q[8], double cvar[8], double ff[8]) {
  double p;
  if (g.cIso > 0) p = q[ID] * g.cIso * g.cIso; else p = q[IP];
  const double entho = 1.0 / (g.gamma0 - 1.0);
  const double d = q[ID], u = q[IU], v = q[IV], w = q[IW], a = q[IA], b = q[IB], c = q[IC];
  const double ecin = 0.5 * (u * u + v * v + w * w) * d;
  const double emag = 0.5 * (a * a + b * b + c * c);
  const double etot = p * entho + ecin + emag;
  const double ptot = p + emag;
  cvar[ID] = d; cvar[IP] = etot; cvar[IU] = d * u; cvar[IV] = d * v; cvar[IW] = d * w;
  cvar[IA] = a; cvar[IB] = b; cvar[IC] = c;
  ff[ID] = d * u;
  ff[IP] = (etot + ptot) * u - a * (a * u + b * v + c * w);
  ff[IU] = d * u * u - a * a + ptot;
  ff[IV] = d * u * v - a * b;
  ff[IW] = d * u * w - a * c;
  ff[IA] = 0.0;
  ff[IB] = b * u - a * v;
  ff[IC] = c * u - a * w;
}

// ---------------------------------------------------------------------------------------------------------
// MHD 1D Riemann solvers (riemann_mhd.h).  NOTE: they overwrite qleft/qright[IA] (and [IP] when isothermal),
// and the callers reuse the modified states (shear correction of flux_y, MHDRunGodunov.cpp:2861-2899).
// ---------------------------------------------------------------------------------------------------------

// riemann_hll (riemann_mhd.h:42-71)
inline void mhd_riemann_hll(const rgpu_params& g, double ql[8], double qr[8], double flux[8]) {
  const double bx_mean = 0.5 * (ql[IA] + qr[IA]);
  ql[IA] = bx_mean; qr[IA] = bx_mean;
  double uL[8], fL[8], uR[8], fR[8];
  find_mhd_flux(g, ql, uL, fL);
  find_mhd_flux(g, qr, uR, fR);
  const double cfl = find_speed_fast<IX>(g, ql), cfr = find_speed_fast<IX>(g, qr);
  const double vl = ql[IU], vr = qr[IU];
  const double sl = fmin(fmin(vl, vr) - fmax(cfl, cfr), 0.0);
  const double sr = fmax(fmax(vl, vr) + fmax(cfl, cfr), 0.0);
  for (int n = 0; n < 8; ++n) flux[n] = (sr * fL[n] - sl * fR[n] + sr * sl * (uR[n] - uL[n])) / (sr - sl);
}

// riemann_llf (riemann_mhd.h:87-118)
inline void mhd_riemann_llf(const rgpu_params& g, double ql[8], double qr[8], double flux[8], double zero_flux = 1.0) {
  const double bx_mean = 0.5 * (ql[IA] + qr[IA]);
  ql[IA] = bx_mean; qr[IA] = bx_mean;
  double uL[8], fL[8], uR[8], fR[8];
  find_mhd_flux(g, ql, uL, fL);
  find_mhd_flux(g, qr, uR, fR);
  for (int n = 0; n < 8; ++n) flux[n] = (ql[n] + qr[n]) / 2 * zero_flux;
  const double cl = find_speed_info_x(g, ql), cr = find_speed_info_x(g, qr);
  const double vel_info = fmax(cl, cr);
  for (int n = 0; n < 8; ++n) flux[n] -= vel_info * (uR[n] - uL[n]) / 2;
}

// riemann_hlld (riemann_mhd.h:140-342)
inline void mhd_riemann_hlld(const rgpu_params& g, double ql[8], double qr[8], double flux[8]) {
  const double entho = 1.0 / (g.gamma0 - 1.0);
  const double a = 0.5 * (ql[IA] + qr[IA]);
  const double sgnm = (a >= 0) ? 1.0 : -1.0;
  ql[IA] = a; qr[IA] = a;
  if (g.cIso > 0) {
    ql[IP] = ql[ID] * g.cIso * g.cIso;
    qr[IP] = qr[ID] * g.cIso * g.cIso;
  }
  const double rl = ql[ID], pl = ql[IP], ul = ql[IU], vl = ql[IV], wl = ql[IW], bl = ql[IB], cl = ql[IC];
  const double ecinl = 0.5 * (ul * ul + vl * vl + wl * wl) * rl;
  const double emagl = 0.5 * (a * a + bl * bl + cl * cl);
  const double etotl = pl * entho + ecinl + emagl;
  const double ptotl = pl + emagl;
  const double vdotbl = ul * a + vl * bl + wl * cl;
  const double rr = qr[ID], pr = qr[IP], ur = qr[IU], vr = qr[IV], wr = qr[IW], br = qr[IB], cr = qr[IC];
  const double ecinr = 0.5 * (ur * ur + vr * vr + wr * wr) * rr;
  const double emagr = 0.5 * (a * a + br * br + cr * cr);
  const double etotr = pr * entho + ecinr + emagr;
  const double ptotr = pr + emagr;
  const double vdotbr = ur * a + vr * br + wr * cr;
  const double cfastl = find_speed_fast<IX>(g, ql), cfastr = find_speed_fast<IX>(g, qr);
  const double sl = fmin(ul, ur) - fmax(cfastl, cfastr);
  const double sr = fmax(ul, ur) + fmax(cfastl, cfastr);
  const double rcl = rl * (ul - sl), rcr = rr * (sr - ur);
  const double ustar = (rcr * ur + rcl * ul + (ptotl - ptotr)) / (rcr + rcl);
  const double ptotstar = (rcr * ptotl + rcl * ptotr + rcl * rcr * (ul - ur)) / (rcr + rcl);
  // left star region
  double estar;
  const double rstarl = rl * (sl - ul) / (sl - ustar);
  estar = rl * (sl - ul) * (sl - ustar) - a * a;
  const double el = rl * (sl - ul) * (sl - ul) - a * a;
  double vstarl, wstarl, bstarl, cstarl;
  if (a * a > 0 && fabs(estar / (a * a) - 1.0) <= 1e-8) {
    vstarl = vl; bstarl = bl; wstarl = wl; cstarl = cl;
  } else {
    vstarl = vl - a * bl * (ustar - ul) / estar;
    bstarl = bl * el / estar;
    wstarl = wl - a * cl * (ustar - ul) / estar;
    cstarl = cl * el / estar;
  }
  const double vdotbstarl = ustar * a + vstarl * bstarl + wstarl * cstarl;
  const double etotstarl = ((sl - ul) * etotl - ptotl * ul + ptotstar * ustar + a * (vdotbl - vdotbstarl)) / (sl - ustar);
  const double sqrrstarl = sqrt(rstarl);
  const double calfvenl = fabs(a) / sqrrstarl;
  const double sal = ustar - calfvenl;
  // right star region
  const double rstarr = rr * (sr - ur) / (sr - ustar);
  estar = rr * (sr - ur) * (sr - ustar) - a * a;
  const double er = rr * (sr - ur) * (sr - ur) - a * a;
  double vstarr, wstarr, bstarr, cstarr;
  if (a * a > 0 && fabs(estar / (a * a) - 1.0) <= 1e-8) {
    vstarr = vr; bstarr = br; wstarr = wr; cstarr = cr;
  } else {
    vstarr = vr - a * br * (ustar - ur) / estar;
    bstarr = br * er / estar;
    wstarr = wr - a * cr * (ustar - ur) / estar;
    cstarr = cr * er / estar;
  }
  const double vdotbstarr = ustar * a + vstarr * bstarr + wstarr * cstarr;
  const double etotstarr = ((sr - ur) * etotr - ptotr * ur + ptotstar * ustar + a * (vdotbr - vdotbstarr)) / (sr - ustar);
  const double sqrrstarr = sqrt(rstarr);
  const double calfvenr = fabs(a) / sqrrstarr;
  const double sar = ustar + calfvenr;
  // double star region
  const double vstarstar = (sqrrstarl * vstarl + sqrrstarr * vstarr + sgnm * (bstarr - bstarl)) / (sqrrstarl + sqrrstarr);
  const double wstarstar = (sqrrstarl * wstarl + sqrrstarr * wstarr + sgnm * (cstarr - cstarl)) / (sqrrstarl + sqrrstarr);
  const double bstarstar = (sqrrstarl * bstarr + sqrrstarr * bstarl + sgnm * sqrrstarl * sqrrstarr * (vstarr - vstarl)) / (sqrrstarl + sqrrstarr);
  const double cstarstar = (sqrrstarl * cstarr + sqrrstarr * cstarl + sgnm * sqrrstarl * sqrrstarr * (wstarr - wstarl)) / (sqrrstarl + sqrrstarr);
  const double vdotbstarstar = ustar * a + vstarstar * bstarstar + wstarstar * cstarstar;
  const double etotstarstarl = etotstarl - sgnm * sqrrstarl * (vdotbstarl - vdotbstarstar);
  const double etotstarstarr = etotstarr + sgnm * sqrrstarr * (vdotbstarr - vdotbstarstar);
  double ro, uo, vo, wo, bo, co, ptoto, etoto, vdotbo;
  if (sl > 0) { ro = rl; uo = ul; vo = vl; wo = wl; bo = bl; co = cl; ptoto = ptotl; etoto = etotl; vdotbo = vdotbl; }
  else if (sal > 0) { ro = rstarl; uo = ustar; vo = vstarl; wo = wstarl; bo = bstarl; co = cstarl; ptoto = ptotstar; etoto = etotstarl; vdotbo = vdotbstarl; }
  else if (ustar > 0) { ro = rstarl; uo = ustar; vo = vstarstar; wo = wstarstar; bo = bstarstar; co = cstarstar; ptoto = ptotstar; etoto = etotstarstarl; vdotbo = vdotbstarstar; }
  else if (sar > 0) { ro = rstarr; uo = ustar; vo = vstarstar; wo = wstarstar; bo = bstarstar; co = cstarstar; ptoto = ptotstar; etoto = etotstarstarr; vdotbo = vdotbstarstar; }
  else if (sr > 0) { ro = rstarr; uo = ustar; vo = vstarr; wo = wstarr; bo = bstarr; co = cstarr; ptoto = ptotstar; etoto = etotstarr; vdotbo = vdotbstarr; }
  else { ro = rr; uo = ur; vo = vr; wo = wr; bo = br; co = cr; ptoto = ptotr; etoto = etotr; vdotbo = vdotbr; }
  flux[ID] = ro * uo;
  flux[IP] = (etoto + ptoto) * uo - a * vdotbo;
  flux[IU] = ro * uo * uo - a * a + ptoto;
  flux[IV] = ro * uo * vo - a * bo;
  flux[IW] = ro * uo * wo - a * co;
  flux[IA] = 0.0;
  flux[IB] = bo * uo - a * vo;
  flux[IC] = co * uo - a * wo;
}

// riemann_mhd dispatch (riemann_mhd.h:355-368): approx / hllc selections compute NOTHING (flux stays as passed)
inline void mhd_riemann(const rgpu_params& g, double ql[8], double qr[8], double flux[8]) {
  if (g.riemannSolver == RGPU_RS_HLL) mhd_riemann_hll(g, ql, qr, flux);
  else if (g.riemannSolver == RGPU_RS_LLF) mhd_riemann_llf(g, ql, qr, flux);
  else if (g.riemannSolver == RGPU_RS_HLLD) mhd_riemann_hlld(g, ql, qr, flux);
}

// ---------------------------------------------------------------------------------------------------------
// 2D magnetic Riemann solver HLLD + compute_emf (riemann_mhd.h:616-821, 1054-1193)
// ---------------------------------------------------------------------------------------------------------
inline double max4(double a0, double a1, double a2, double a3) {
  double r = a0; r = (a1 > r) ? a1 : r; r = (a2 > r) ? a2 : r; r = (a3 > r) ? a3 : r; return r;
}
inline double min4(double a0, double a1, double a2, double a3) {
  double r = a0; r = (a1 < r) ? a1 : r; r = (a2 < r) ? a2 : r; r = (a3 < r) ? a3 : r; return r;
}
inline double max5(double a0, double a1, double a2, double a3, double a4) {
  double r = a0; r = (a1 > r) ? a1 : r; r = (a2 > r) ? a2 : r; r = (a3 > r) ? a3 : r; r = (a4 > r) ? a4 : r; return r;
}

// find_speed_alfven(d, a) (mhd_utils.h:82-88)
inline double find_speed_alfven(double d, double a) { return sqrt(a * a / d); }

// the HLL average shared by mag_riemann2d_hlla / hllf (riemann_mhd.h:449-453, 497-501)
inline double mag_hll_average(const double qLLRR[4][8], const double eLLRR[4], double SL, double SR, double SB, double ST) {
  const double* qLL = qLLRR[0]; const double* qRR = qLLRR[3];
  const double ELL = eLLRR[0], ERL = eLLRR[1], ELR = eLLRR[2], ERR = eLLRR[3];
  return (SL * SB * ERR - SL * ST * ERL - SR * SB * ELR + SR * ST * ELL) / (SR - SL) / (ST - SB)
         - ST * SB / (ST - SB) * (qRR[IA] - qLL[IA])
         + SR * SL / (SR - SL) * (qRR[IB] - qLL[IB]);
}

// mag_riemann2d_hlla (riemann_mhd.h:417-457): Alfven speeds, floor smallc
inline double mag_riemann2d_hlla(const rgpu_params& g, const double qLLRR[4][8], const double eLLRR[4]) {
  const double* qLL = qLLRR[0]; const double* qRL = qLLRR[1]; const double* qLR = qLLRR[2]; const double* qRR = qLLRR[3];
  const double cMaxx = max5(find_speed_alfven(qLL[ID], qLL[IA]), find_speed_alfven(qLR[ID], qLR[IA]),
                            find_speed_alfven(qRL[ID], qRL[IA]), find_speed_alfven(qRR[ID], qRR[IA]), g.smallc);
  const double cMaxy = max5(find_speed_alfven(qLL[ID], qLL[IB]), find_speed_alfven(qLR[ID], qLR[IB]),
                            find_speed_alfven(qRL[ID], qRL[IB]), find_speed_alfven(qRR[ID], qRR[IB]), g.smallc);
  const double SL = fmin(min4(qLL[IU], qLR[IU], qRL[IU], qRR[IU]) - cMaxx, 0.0);
  const double SR = fmax(max4(qLL[IU], qLR[IU], qRL[IU], qRR[IU]) + cMaxx, 0.0);
  const double SB = fmin(min4(qLL[IV], qLR[IV], qRL[IV], qRR[IV]) - cMaxy, 0.0);
  const double ST = fmax(max4(qLL[IV], qLR[IV], qRL[IV], qRR[IV]) + cMaxy, 0.0);
  return mag_hll_average(qLLRR, eLLRR, SL, SR, SB, ST);
}

// mag_riemann2d_hllf (riemann_mhd.h:463-505): fast magnetosonic speeds
inline double mag_riemann2d_hllf(const rgpu_params& g, const double qLLRR[4][8], const double eLLRR[4]) {
  const double* qLL = qLLRR[0]; const double* qRL = qLLRR[1]; const double* qLR = qLLRR[2]; const double* qRR = qLLRR[3];
  const double cMaxx = max4(find_speed_fast<IX>(g, qLL), find_speed_fast<IX>(g, qLR), find_speed_fast<IX>(g, qRL), find_speed_fast<IX>(g, qRR));
  const double cMaxy = max4(find_speed_fast<IY>(g, qLL), find_speed_fast<IY>(g, qLR), find_speed_fast<IY>(g, qRL), find_speed_fast<IY>(g, qRR));
  const double SL = fmin(min4(qLL[IU], qLR[IU], qRL[IU], qRR[IU]) - cMaxx, 0.0);
  const double SR = fmax(max4(qLL[IU], qLR[IU], qRL[IU], qRR[IU]) + cMaxx, 0.0);
  const double SB = fmin(min4(qLL[IV], qLR[IV], qRL[IV], qRR[IV]) - cMaxy, 0.0);
  const double ST = fmax(max4(qLL[IV], qLR[IV], qRL[IV], qRR[IV]) + cMaxy, 0.0);
  return mag_hll_average(qLLRR, eLLRR, SL, SR, SB, ST);
}

// mag_riemann2d_llf (riemann_mhd.h:517-609): mean of the four corner E + two 1D LLF solves on face-averaged states
inline double mag_riemann2d_llf(const rgpu_params& g, const double qLLRR[4][8], const double eLLRR[4]) {
  const double* qLL = qLLRR[0]; const double* qRL = qLLRR[1]; const double* qLR = qLLRR[2]; const double* qRR = qLLRR[3];
  double E = (eLLRR[0] + eLLRR[1] + eLLRR[2] + eLLRR[3]) / 4;
  double ql[8], qr[8], fx[8], fy[8];
  for (int n = 0; n < 8; ++n) { ql[n] = (qLL[n] + qLR[n]) / 2; qr[n] = (qRR[n] + qRL[n]) / 2; }
  mhd_riemann_llf(g, ql, qr, fx, 0.0);
  static const int swap[8] = {ID, IP, IV, IU, IW, IB, IA, IC};   // the y problem: u<->v, a<->b
  for (int n = 0; n < 8; ++n) { ql[n] = (qLL[swap[n]] + qRL[swap[n]]) / 2; qr[n] = (qRR[swap[n]] + qLR[swap[n]]) / 2; }
  mhd_riemann_llf(g, ql, qr, fy, 0.0);
  E += (fx[IB] - fy[IB]);
  return E;
}

// slope_type == 3, positivity preserving slopes: the limiter of slope_unsplit_hydro_2d / _3d
// (slope_mhd.h:131-168, 336-407).  nb = 9 (2D) or 27 (3D) neighbourhood values INCLUDING the centre value qc;
// d[] = centred half differences per direction; returns the common factor dlim (dq[dir] = dlim * d[dir])
inline double positivity_limiter(const double* nb, int count, double qc, const double* d, int ndim) {
  double vmin = nb[0] - qc, vmax = nb[0] - qc;
  for (int n = 1; n < count; ++n) {
    const double df = nb[n] - qc;
    vmin = (df < vmin) ? df : vmin;
    vmax = (df > vmax) ? df : vmax;
  }
  double dff = 0.0;
  for (int n = 0; n < ndim; ++n) dff = dff + fabs(d[n]);     // FABS(dfx) + FABS(dfy) [+ FABS(dfz)], left to right
  dff = 0.5 * dff;
  if (dff > 0.0) return fmin(1.0, fmin(fabs(vmin), fabs(vmax)) / dff);
  return 1.0;
}

// qLLRR order: ILL=0, IRL=1, ILR=2, IRR=3 (constants.h:176-181)
inline double mag_riemann2d_hlld(const rgpu_params& g, const double qLLRR[4][8], const double eLLRR[4]) {
  const double* qLL = qLLRR[0]; const double* qRL = qLLRR[1]; const double* qLR = qLLRR[2]; const double* qRR = qLLRR[3];
  const double ELL = eLLRR[0], ERL = eLLRR[1], ELR = eLLRR[2], ERR = eLLRR[3];
  const double rLL = qLL[ID], pLL = qLL[IP], uLL = qLL[IU], vLL = qLL[IV], aLL = qLL[IA], bLL = qLL[IB], cLL = qLL[IC];
  const double rLR = qLR[ID], pLR = qLR[IP], uLR = qLR[IU], vLR = qLR[IV], aLR = qLR[IA], bLR = qLR[IB], cLR = qLR[IC];
  const double rRL = qRL[ID], pRL = qRL[IP], uRL = qRL[IU], vRL = qRL[IV], aRL = qRL[IA], bRL = qRL[IB], cRL = qRL[IC];
  const double rRR = qRR[ID], pRR = qRR[IP], uRR = qRR[IU], vRR = qRR[IV], aRR = qRR[IA], bRR = qRR[IB], cRR = qRR[IC];
  const double cFastLLx = find_speed_fast<IX>(g, qLL), cFastLRx = find_speed_fast<IX>(g, qLR);
  const double cFastRLx = find_speed_fast<IX>(g, qRL), cFastRRx = find_speed_fast<IX>(g, qRR);
  const double cFastLLy = find_speed_fast<IY>(g, qLL), cFastLRy = find_speed_fast<IY>(g, qLR);
  const double cFastRLy = find_speed_fast<IY>(g, qRL), cFastRRy = find_speed_fast<IY>(g, qRR);
  const double SL = min4(uLL, uLR, uRL, uRR) - max4(cFastLLx, cFastLRx, cFastRLx, cFastRRx);
  const double SR = max4(uLL, uLR, uRL, uRR) + max4(cFastLLx, cFastLRx, cFastRLx, cFastRRx);
  const double SB = min4(vLL, vLR, vRL, vRR) - max4(cFastLLy, cFastLRy, cFastRLy, cFastRRy);
  const double ST = max4(vLL, vLR, vRL, vRR) + max4(cFastLLy, cFastLRy, cFastRLy, cFastRRy);
  const double PtotLL = pLL + 0.5 * (aLL * aLL + bLL * bLL + cLL * cLL);
  const double PtotLR = pLR + 0.5 * (aLR * aLR + bLR * bLR + cLR * cLR);
  const double PtotRL = pRL + 0.5 * (aRL * aRL + bRL * bRL + cRL * cRL);
  const double PtotRR = pRR + 0.5 * (aRR * aRR + bRR * bRR + cRR * cRR);
  const double rcLLx = rLL * (uLL - SL), rcRLx = rRL * (SR - uRL);
  const double rcLRx = rLR * (uLR - SL), rcRRx = rRR * (SR - uRR);
  const double rcLLy = rLL * (vLL - SB), rcLRy = rLR * (ST - vLR);
  const double rcRLy = rRL * (vRL - SB), rcRRy = rRR * (ST - vRR);
  const double ustar = (rcLLx * uLL + rcLRx * uLR + rcRLx * uRL + rcRRx * uRR + (PtotLL - PtotRL + PtotLR - PtotRR)) /
                       (rcLLx + rcLRx + rcRLx + rcRRx);
  const double vstar = (rcLLy * vLL + rcLRy * vLR + rcRLy * vRL + rcRRy * vRR + (PtotLL - PtotLR + PtotRL - PtotRR)) /
                       (rcLLy + rcLRy + rcRLy + rcRRy);
  const double rstarLLx = rLL * (SL - uLL) / (SL - ustar);
  const double BstarLL = bLL * (SL - uLL) / (SL - ustar);
  const double rstarLLy = rLL * (SB - vLL) / (SB - vstar);
  const double AstarLL = aLL * (SB - vLL) / (SB - vstar);
  const double rstarLL = rLL * (SL - uLL) / (SL - ustar) * (SB - vLL) / (SB - vstar);
  const double EstarLLx = ustar * BstarLL - vLL * aLL;
  const double EstarLLy = uLL * bLL - vstar * AstarLL;
  const double EstarLL = ustar * BstarLL - vstar * AstarLL;
  const double rstarLRx = rLR * (SL - uLR) / (SL - ustar);
  const double BstarLR = bLR * (SL - uLR) / (SL - ustar);
  const double rstarLRy = rLR * (ST - vLR) / (ST - vstar);
  const double AstarLR = aLR * (ST - vLR) / (ST - vstar);
  const double rstarLR = rLR * (SL - uLR) / (SL - ustar) * (ST - vLR) / (ST - vstar);
  const double EstarLRx = ustar * BstarLR - vLR * aLR;
  const double EstarLRy = uLR * bLR - vstar * AstarLR;
  const double EstarLR = ustar * BstarLR - vstar * AstarLR;
  const double rstarRLx = rRL * (SR - uRL) / (SR - ustar);
  const double BstarRL = bRL * (SR - uRL) / (SR - ustar);
  const double rstarRLy = rRL * (SB - vRL) / (SB - vstar);
  const double AstarRL = aRL * (SB - vRL) / (SB - vstar);
  const double rstarRL = rRL * (SR - uRL) / (SR - ustar) * (SB - vRL) / (SB - vstar);
  const double EstarRLx = ustar * BstarRL - vRL * aRL;
  const double EstarRLy = uRL * bRL - vstar * AstarRL;
  const double EstarRL = ustar * BstarRL - vstar * AstarRL;
  const double rstarRRx = rRR * (SR - uRR) / (SR - ustar);
  const double BstarRR = bRR * (SR - uRR) / (SR - ustar);
  const double rstarRRy = rRR * (ST - vRR) / (ST - vstar);
  const double AstarRR = aRR * (ST - vRR) / (ST - vstar);
  const double rstarRR = rRR * (SR - uRR) / (SR - ustar) * (ST - vRR) / (ST - vstar);
  const double EstarRRx = ustar * BstarRR - vRR * aRR;
  const double EstarRRy = uRR * bRR - vstar * AstarRR;
  const double EstarRR = ustar * BstarRR - vstar * AstarRR;
  const double calfvenL = max5(fabs(aLR) / sqrt(rstarLRx), fabs(AstarLR) / sqrt(rstarLR), fabs(aLL) / sqrt(rstarLLx),
                               fabs(AstarLL) / sqrt(rstarLL), g.smallc);
  const double calfvenR = max5(fabs(aRR) / sqrt(rstarRRx), fabs(AstarRR) / sqrt(rstarRR), fabs(aRL) / sqrt(rstarRLx),
                               fabs(AstarRL) / sqrt(rstarRL), g.smallc);
  const double calfvenB = max5(fabs(bLL) / sqrt(rstarLLy), fabs(BstarLL) / sqrt(rstarLL), fabs(bRL) / sqrt(rstarRLy),
                               fabs(BstarRL) / sqrt(rstarRL), g.smallc);
  const double calfvenT = max5(fabs(bLR) / sqrt(rstarLRy), fabs(BstarLR) / sqrt(rstarLR), fabs(bRR) / sqrt(rstarRRy),
                               fabs(BstarRR) / sqrt(rstarRR), g.smallc);
  const double SAL = fmin(ustar - calfvenL, 0.0);
  const double SAR = fmax(ustar + calfvenR, 0.0);
  const double SAB = fmin(vstar - calfvenB, 0.0);
  const double SAT = fmax(vstar + calfvenT, 0.0);
  const double AstarT = (SAR * AstarRR - SAL * AstarLR) / (SAR - SAL);
  const double AstarB = (SAR * AstarRL - SAL * AstarLL) / (SAR - SAL);
  const double BstarR = (SAT * BstarRR - SAB * BstarRL) / (SAT - SAB);
  const double BstarL = (SAT * BstarLR - SAB * BstarLL) / (SAT - SAB);
  double E = 0, tmpE = 0;
  // "sort of boolean": pos iff the sign bit is clear (riemann_mhd.h:759-762)
  const int SB_pos = (int)(1 + copysign(1.0, SB)) / 2, SB_neg = 1 - SB_pos;
  const int ST_pos = (int)(1 + copysign(1.0, ST)) / 2, ST_neg = 1 - ST_pos;
  const int SL_pos = (int)(1 + copysign(1.0, SL)) / 2, SL_neg = 1 - SL_pos;
  const int SR_pos = (int)(1 + copysign(1.0, SR)) / 2, SR_neg = 1 - SR_pos;
  tmpE = (SAL * SAB * EstarRR - SAL * SAT * EstarRL - SAR * SAB * EstarLR + SAR * SAT * EstarLL) / (SAR - SAL) / (SAT - SAB) -
         SAT * SAB / (SAT - SAB) * (AstarT - AstarB) + SAR * SAL / (SAR - SAL) * (BstarR - BstarL);
  E += (SB_neg * ST_pos * SL_neg * SR_pos) * tmpE;
  tmpE = (SAR * EstarLLx - SAL * EstarRLx + SAR * SAL * (bRL - bLL)) / (SAR - SAL);
  tmpE = SL_pos * ELL + SL_neg * SR_neg * ERL + SL_neg * SR_pos * tmpE;
  E += SB_pos * tmpE;
  tmpE = (SAR * EstarLRx - SAL * EstarRRx + SAR * SAL * (bRR - bLR)) / (SAR - SAL);
  tmpE = SL_pos * ELR + SL_neg * SR_neg * ERR + SL_neg * SR_pos * tmpE;
  E += (SB_neg * ST_neg) * tmpE;
  tmpE = (SAT * EstarLLy - SAB * EstarLRy - SAT * SAB * (aLR - aLL)) / (SAT - SAB);
  E += (SB_neg * ST_pos * SL_pos) * tmpE;
  tmpE = (SAT * EstarRLy - SAB * EstarRRy - SAT * SAB * (aRR - aRL)) / (SAT - SAB);
  E += (SB_neg * ST_pos * SL_neg * SR_neg) * tmpE;
  return E;
}

// compute_emf<emfDir> (riemann_mhd.h:1054-1193); qEdge order IRT=0, IRB=1, ILT=2, ILB=3 (constants.h:168-173);
// EMFDIR: 0 = EMFX, 1 = EMFY, 2 = EMFZ (constants.h:184-188)
template <int EMFDIR>
inline double compute_emf(const rgpu_params& g, const double qEdge[4][8], double xPos = 0) {
  const double* qRT = qEdge[0]; const double* qRB = qEdge[1]; const double* qLT = qEdge[2]; const double* qLB = qEdge[3];
  double qLLRR[4][8];
  double* qLL = qLLRR[0]; double* qRL = qLLRR[1]; double* qLR = qLLRR[2]; double* qRR = qLLRR[3];
  qLL[ID] = qRT[ID]; qRL[ID] = qLT[ID]; qLR[ID] = qRB[ID]; qRR[ID] = qLB[ID];
  const double cIso = g.cIso;
  if (cIso > 0) {
    qLL[IP] = qLL[ID] * cIso * cIso; qRL[IP] = qRL[ID] * cIso * cIso;
    qLR[IP] = qLR[ID] * cIso * cIso; qRR[IP] = qRR[ID] * cIso * cIso;
  } else {
    qLL[IP] = qRT[IP]; qRL[IP] = qLT[IP]; qLR[IP] = qRB[IP]; qRR[IP] = qLB[IP];
  }
  int iu, iv, iw, ia, ib, ic;
  if (EMFDIR == 2) { iu = IU; iv = IV; iw = IW; ia = IA; ib = IB; ic = IC; }
  else if (EMFDIR == 1) { iu = IW; iv = IU; iw = IV; ia = IC; ib = IA; ic = IB; }
  else { iu = IV; iv = IW; iw = IU; ia = IB; ib = IC; ic = IA; }
  qLL[IU] = qRT[iu]; qRL[IU] = qLT[iu]; qLR[IU] = qRB[iu]; qRR[IU] = qLB[iu];
  qLL[IV] = qRT[iv]; qRL[IV] = qLT[iv]; qLR[IV] = qRB[iv]; qRR[IV] = qLB[iv];
  qLL[IA] = 0.5 * (qRT[ia] + qLT[ia]);
  qRL[IA] = 0.5 * (qRT[ia] + qLT[ia]);
  qLR[IA] = 0.5 * (qRB[ia] + qLB[ia]);
  qRR[IA] = 0.5 * (qRB[ia] + qLB[ia]);
  qLL[IB] = 0.5 * (qRT[ib] + qRB[ib]);
  qRL[IB] = 0.5 * (qLT[ib] + qLB[ib]);
  qLR[IB] = 0.5 * (qRT[ib] + qRB[ib]);
  qRR[IB] = 0.5 * (qLT[ib] + qLB[ib]);
  qLL[IW] = qRT[iw]; qRL[IW] = qLT[iw]; qLR[IW] = qRB[iw]; qRR[IW] = qLB[iw];
  qLL[IC] = qRT[ic]; qRL[IC] = qLT[ic]; qLR[IC] = qRB[ic]; qRR[IC] = qLB[ic];
  double eLLRR[4];
  eLLRR[0] = qLL[IU] * qLL[IB] - qLL[IV] * qLL[IA];
  eLLRR[1] = qRL[IU] * qRL[IB] - qRL[IV] * qRL[IA];
  eLLRR[2] = qLR[IU] * qLR[IB] - qLR[IV] * qLR[IA];
  eLLRR[3] = qRR[IU] * qRR[IB] - qRR[IV] * qRR[IA];
  double emf = 0;
  if (g.magRiemannSolver == RGPU_MAG_HLLD) emf = mag_riemann2d_hlld(g, qLLRR, eLLRR);
  else if (g.magRiemannSolver == RGPU_MAG_HLLA) emf = mag_riemann2d_hlla(g, qLLRR, eLLRR);
  else if (g.magRiemannSolver == RGPU_MAG_HLLF) emf = mag_riemann2d_hllf(g, qLLRR, eLLRR);
  else if (g.magRiemannSolver == RGPU_MAG_LLF) emf = mag_riemann2d_llf(g, qLLRR, eLLRR);
  if (g.Omega0 > 0) {
    if (EMFDIR == 0) {
      const double shear = -1.5 * g.Omega0 * xPos;
      if (shear > 0) emf += shear * qLL[IB]; else emf += shear * qRR[IB];
    }
    if (EMFDIR == 2) {
      const double shear = -1.5 * g.Omega0 * (xPos - g.dx / 2);
      if (shear > 0) emf -= shear * qLL[IA]; else emf -= shear * qRR[IA];
    }
  }
  return emf;
}

}  // namespace orc
