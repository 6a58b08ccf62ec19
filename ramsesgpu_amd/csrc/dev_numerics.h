// dev_numerics.h -- per-cell numerics of the unsplit Godunov / MUSCL-Hancock step as gfx950 device functions.
//
// Everything here is straight-line fp64 VALU code (this path has no GEMM shape; MFMA is not used).  The
// arithmetic follows the reference's operand order expression by expression, because the parity bar is
// bit-identity with euler_cpu: with -ffp-contract=off, IEEE fp64 divide / sqrt and v_max/v_min_f64, every
// value computed here equals the CPU value.  Each function cites the reference routine it stands in for.
//
// RG_DEVFN is supplied by the backend header (rg_backend.h): __device__ __forceinline__ for hipcc.
#pragma once
#include "rg_backend.h"

namespace rgpu_dev {

using rgpu::rg_recip_t;
using rgpu::rg_recip;
using rgpu::rg_recip2;
using rgpu::rg_recip4;
using rgpu::rg_div;
using rgpu::rg_sqrt;
using rgpu::rg_sqrt_pos;
using rgpu::rg_recip_sqrt_pos;

enum { ID = 0, IP = 1, IU = 2, IV = 3, IW = 4, IA = 5, IB = 6, IC = 7 };
enum { XD = 0, YD = 1, ZD = 2 };

// Kernel-argument block (passed by value; lives in SGPRs / kernarg segment).  Derived from rgpu_params.
struct DevParams {
  int isize, jsize, ksize, gw, nx, ny, nz, nvar;
  int three_d, mhd, rot, shearbox;
  int dirwise_update;          // hydro unsplitVersion 2: fluxes applied direction by direction
  int xcd_sub;                 // host side only: sub-band size (cells) of the XCD-aware workgroup order, 0 = linear
  int zlo_copy, zhi_copy;      // z faces that are slab interfaces (RGPU_BC_COPY): the neighbour's cells continue there
  unsigned sj, sk;             // flat strides of +1 in j and k
  unsigned long long ncell;    // component stride
  // F / emf of the 3D MHD step (scratch, not part of the ABI) have a pitch of their own: rows of fsj doubles, a multiple of 16, and
  // the first interior cell of every row (i = gw) on a 128-byte boundary (foff), so that the 16-cell row segments a wave of the sweep
  // writes are whole lines (rows of isize = nx + 6 doubles start 24 B off a granule: 1.3-1.45 x write amplification, rounds 3-5).
  // Every other solver family: the state's own layout (fsj = sj, fsk = sk, foff = 0, fN = ncell).
  unsigned fsj, fsk, foff;
  unsigned long long fN;       // component stride of F / emf
  double dx, dy, dz, xMin, deltaX;   // deltaX = xMax - xMin
  double gamma0, cIso, smallr, smallc, smallp, smallpp, gamma6, Omega0;
  double slope_type, mag_slope_type;
  int niter_riemann, riemannSolver, magRiemannSolver, grav_on;
  // static gravity: (0.5 * dt) * g of the CURRENT step, set by the step driver before it launches (the reference's
  // "HALF_F * dt * h_gravity(i,j,k,d)" with the uniform field its problems fill in)
  double hgx, hgy, hgz;
  // grav_on == 2: per-cell field G[d * ncell + idx] (the reference's h_gravity / d_gravity, uploaded once with
  // rgpu_set_gravity_field) and hdt = 0.5 * dt of the current step
  const double* G;
  double hdt;
};

// (0.5 * dt) * g at cell m.  GFIELD is a compile-time switch (kernels are instantiated twice and chosen at launch): a
// run-time test here would put a branch with loads into every traced state and cut the Riemann kernels' straight-line
// code into pieces (measured: 35.9 -> 47.3 ms for the 3D MHD Riemann kernel at 512^3 with the branch never taken).
template <bool GFIELD>
RG_DEVFN void half_dt_gravity(const DevParams& g, unsigned m, double& gx, double& gy, double& gz) {
  if (GFIELD) {
    gx = g.hdt * g.G[m];
    gy = g.hdt * g.G[m + g.ncell];
    gz = g.three_d ? g.hdt * g.G[m + 2 * g.ncell] : 0.0;
  } else {
    gx = g.hgx; gy = g.hgy; gz = g.hgz;
  }
}

// flat index of cell (i, j, k) in F / emf
RG_DEVFN unsigned flux_index(const DevParams& g, int i, int j, int k) { return g.foff + (unsigned)i + (unsigned)j * g.fsj + (unsigned)k * g.fsk; }

struct Prim8 {  // primitive MHD state in some frame: density, pressure, 3 velocities, 3 field components
  double r, p, u, v, w, a, b, c;
};

// ---------------------------------------------------------------------------------------------------------
// slopes
// ---------------------------------------------------------------------------------------------------------

// limiter of slope_unsplit_hydro_{2d,3d}, slope_unsplit_3d (type 2), slope_unsplit_mhd_{2d,3d}
// (slope.h:118-145,386-425; slope_mhd.h:105-128,463-497,549-571,636-700)
RG_DEVFN double tvd_slope(double st, double qm, double q0, double qp) {
  const double dlft = st * (q0 - qm);
  const double drgt = st * (qp - q0);
  const double dcen = 0.5 * (qp - qm);
  const double dsgn = (dcen >= 0.0) ? 1.0 : -1.0;
  const double slop = fmin(fabs(dlft), fabs(drgt));
  const double dlim = ((dlft * drgt) <= 0.0) ? 0.0 : slop;
  return dsgn * fmin(dlim, fabs(dcen));
}

// HALF of that slope, as the 3D MHD trace consumes it (trace_mhd.h:1902-1936: "0.5 * dq"), formed without a compare or a select:
//   0.5 * dsgn * min(min(|st dl|, |st dr|) [dl dr > 0], |0.5 dc|)  =  max(min(a, b, c), 0) + min(max(a, b, c), 0)
// with a = (st / 2) dl, b = (st / 2) dr, c = dc / 4.  Scaling by 1/2 commutes with every rounding, the minimum and the product's sign,
// and dl, dr > 0 imply dc > 0 (qp > q0 > qm), so where all three have one sign the sum is that sign's smallest magnitude -- the
// reference's value bit for bit -- and +-0 elsewhere: 11 fp64 instructions instead of 16 with two compares and three selects.
// Differences to tvd_slope: the sign of a zero result, and |st dl * st dr| < 2^-1074 (the reference's product underflows to 0 and
// its test fails: it returns 0, this returns the sub-1e-150 minimum).
RG_DEVFN double tvd_half_slope(double st, double qm, double q0, double qp) {
  const double hs = 0.5 * st;
  const double a = hs * (q0 - qm), b = hs * (qp - q0), c = 0.25 * (qp - qm);
  const double lo = fmin(fmin(a, b), c), hi = fmax(fmax(a, b), c);
  return fmax(lo, 0.0) + fmin(hi, 0.0);
}

// slope_unsplit_3d, slope_type == 1 (slope.h:351-384)
RG_DEVFN double minmod_slope(double qm, double q0, double qp) {
  const double dlft = q0 - qm;
  const double drgt = qp - q0;
  if ((dlft * drgt) <= 0.0) return 0.0;
  return (dlft > 0) ? fmin(dlft, drgt) : fmax(dlft, drgt);
}

// HALF of minmod_slope, compare-free: max(min(a, b), 0) + min(max(a, b), 0) with a = dlft / 2, b = drgt / 2 (same remarks as tvd_half_slope)
RG_DEVFN double minmod_half_slope(double qm, double q0, double qp) {
  const double a = 0.5 * (q0 - qm), b = 0.5 * (qp - q0);
  return fmax(fmin(a, b), 0.0) + fmin(fmax(a, b), 0.0);
}

// slope_type 3, positivity preserving slopes of slope_unsplit_hydro_2d / _3d (slope_mhd.h:131-168, 336-407):
// every centred half difference d of a variable is scaled by min(1, min(|vmin|,|vmax|) / (0.5 * sum|d|)), vmin / vmax
// being the extreme differences to the 3x3(x3) neighbourhood.  lo / hi are the extreme neighbourhood VALUES
// (centre included): subtraction is monotonic, so min_i fl(q_i - qc) == fl(min_i q_i - qc) bit for bit.
RG_DEVFN double positivity_limiter(double lo, double hi, double qc, double sum_abs_d) {
  const double dff = 0.5 * sum_abs_d;
  if (dff > 0.0) return fmin(1.0, fmin(fabs(lo - qc), fabs(hi - qc)) / dff);
  return 1.0;
}

// ---------------------------------------------------------------------------------------------------------
// hydro: equation of state, Riemann solvers
// ---------------------------------------------------------------------------------------------------------

// constoprim_2D/3D + eos (constoprim.h:29-111).  q = {r,p,u,v,w}; returns the sound speed
template <int NV>
RG_DEVFN double hydro_prim(const DevParams& g, const double* u, double* q) {
  q[ID] = fmax(u[ID], g.smallr);
  const rg_recip_t inv_r = rg_recip(q[ID]);
  q[IU] = rg_div(u[IU], inv_r);
  q[IV] = rg_div(u[IV], inv_r);
  if (NV == 5) q[IW] = rg_div(u[IW], inv_r);
  double eken;
  if (NV == 5) eken = 0.5 * (q[IU] * q[IU] + q[IV] * q[IV] + q[IW] * q[IW]);
  else eken = 0.5 * (q[IU] * q[IU] + q[IV] * q[IV]);
  if (g.cIso > 0) {
    q[IP] = q[ID] * g.cIso * g.cIso;
    return g.cIso;
  }
  const double eint = rg_div(u[IP], inv_r) - eken;
  q[IP] = fmax((g.gamma0 - 1.0) * q[ID] * eint, q[ID] * g.smallp);
  return rg_sqrt(rg_div(g.gamma0 * q[IP], inv_r));
}

// cmpflx (cmpflx.h:21-48)
template <int NV>
RG_DEVFN void hydro_flux_from_state(const DevParams& g, const double* qg, double* flux) {
  flux[ID] = qg[ID] * qg[IU];
  flux[IU] = flux[ID] * qg[IU] + qg[IP];
  flux[IV] = flux[ID] * qg[IV];
  if (NV == 5) flux[IW] = flux[ID] * qg[IW];
  const double entho = 1.0 / (g.gamma0 - 1.0);
  double ekin;
  if (NV == 5) ekin = 0.5 * qg[ID] * (qg[IU] * qg[IU] + qg[IV] * qg[IV] + qg[IW] * qg[IW]);
  else ekin = 0.5 * qg[ID] * (qg[IU] * qg[IU] + qg[IV] * qg[IV]);
  const double etot = qg[IP] * entho + ekin;
  flux[IP] = qg[IU] * (etot + qg[IP]);
}

// saturate_cpu (gpu_macros.cpp:25-30): clamp THROUGH FLOAT
RG_DEVFN double saturate_via_float(double x) {
  const float a = (float)x;
  if (a != a) return 0.0;
  return (double)(a >= 1.0f ? 1.0f : a <= 0.0f ? 0.0f : a);
}

// riemann_approx (riemann.h:29-159)
template <int NV>
RG_DEVFN void riemann_approx(const DevParams& g, const double* ql, const double* qr, double* flux) {
  const double rl = fmax(ql[ID], g.smallr), ul = ql[IU], pl = fmax(ql[IP], rl * g.smallp);
  const double rr = fmax(qr[ID], g.smallr), ur = qr[IU], pr = fmax(qr[IP], rr * g.smallp);
  const double cl = g.gamma0 * pl * rl, cr = g.gamma0 * pr * rr;
  double wl = rg_sqrt(cl), wr = rg_sqrt(cr);
  double pstar = fmax(rg_div((wr * pl + wl * pr) + wl * wr * (ul - ur), rg_recip(wl + wr)), 0.0);
  const rg_recip_t inv_pl = rg_recip(pl), inv_pr = rg_recip(pr);   // reused by every Newton iteration
  double pold = pstar, conv = 1.0;
  for (int iter = 0; iter < g.niter_riemann && conv > 1e-6; ++iter) {
    const double wwl = rg_sqrt(cl * (1.0 + rg_div(g.gamma6 * (pold - pl), inv_pl)));
    const double wwr = rg_sqrt(cr * (1.0 + rg_div(g.gamma6 * (pold - pr), inv_pr)));
    const double q_l = rg_div(2.0 * wwl * wwl * wwl, rg_recip(wwl * wwl + cl));
    const double q_r = rg_div(2.0 * wwr * wwr * wwr, rg_recip(wwr * wwr + cr));
    const double usl = ul - rg_div(pold - pl, rg_recip(wwl));
    const double usr = ur + rg_div(pold - pr, rg_recip(wwr));
    const double delp = fmax(rg_div(q_r * q_l, rg_recip(q_r + q_l)) * (usl - usr), -pold);
    pold = pold + delp;
    conv = fabs(rg_div(delp, rg_recip(pold + g.smallpp)));
  }
  pstar = pold;
  wl = rg_sqrt(cl * (1.0 + rg_div(g.gamma6 * (pstar - pl), inv_pl)));
  wr = rg_sqrt(cr * (1.0 + rg_div(g.gamma6 * (pstar - pr), inv_pr)));
  const double ustar = 0.5 * (ul + rg_div(pl - pstar, rg_recip(wl)) + ur - rg_div(pr - pstar, rg_recip(wr)));
  const double sgnm = copysign(1.0, ustar);
  const bool from_left = sgnm > 0.0;
  const double ro = from_left ? rl : rr, uo = from_left ? ul : ur, po = from_left ? pl : pr, wo = from_left ? wl : wr;
  const rg_recip_t inv_ro = rg_recip(ro);
  const double co = fmax(g.smallc, rg_sqrt(fabs(rg_div(g.gamma0 * po, inv_ro))));
  const double rstar = fmax(rg_div(ro, rg_recip(1.0 + rg_div(ro * (po - pstar), rg_recip(wo * wo)))), g.smallr);
  const double cstar = fmax(g.smallc, rg_sqrt(fabs(rg_div(g.gamma0 * pstar, rg_recip(rstar)))));
  double spout = co - sgnm * uo;
  double spin = cstar - sgnm * ustar;
  const double ushock = rg_div(wo, inv_ro) - sgnm * uo;
  if (pstar >= po) { spin = ushock; spout = ushock; }
  const double scr = fmax(spout - spin, g.smallc + fabs(spout + spin));
  double frac = 0.5 * (1.0 + rg_div(spout + spin, rg_recip(scr)));
  frac = (frac != frac) ? 0.0 : saturate_via_float(frac);
  double qg[NV];
  qg[ID] = frac * rstar + (1.0 - frac) * ro;
  qg[IU] = frac * ustar + (1.0 - frac) * uo;
  qg[IP] = frac * pstar + (1.0 - frac) * po;
  if (spout < 0.0) { qg[ID] = ro; qg[IU] = uo; qg[IP] = po; }
  if (spin > 0.0) { qg[ID] = rstar; qg[IU] = ustar; qg[IP] = pstar; }
  qg[IV] = from_left ? ql[IV] : qr[IV];
  if (NV == 5) qg[IW] = from_left ? ql[IW] : qr[IW];
  hydro_flux_from_state<NV>(g, qg, flux);
}

// riemann_hll (riemann.h:175-255)
template <int NV>
RG_DEVFN void riemann_hll(const DevParams& g, const double* ql, const double* qr, double* flux) {
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
#pragma unroll
  for (int n = 0; n < NV; ++n) flux[n] = (SR * fL[n] - SL * fR[n] + SR * SL * (uR[n] - uL[n])) / (SR - SL);
}

// riemann_hllc (riemann.h:269-371)
template <int NV>
RG_DEVFN void riemann_hllc(const DevParams& g, const double* ql, const double* qr, double* flux) {
  const double entho = 1.0 / (g.gamma0 - 1.0);
  const double rl = fmax(ql[ID], g.smallr), pl = fmax(ql[IP], rl * g.smallp), ul = ql[IU];
  double ecinl = 0.5 * rl * ul * ul;
  ecinl += 0.5 * rl * ql[IV] * ql[IV];
  if (NV == 5) ecinl += 0.5 * rl * ql[IW] * ql[IW];
  const double etotl = pl * entho + ecinl;
  const double rr = fmax(qr[ID], g.smallr), pr = fmax(qr[IP], rr * g.smallp), ur = qr[IU];
  double ecinr = 0.5 * rr * ur * ur;
  ecinr += 0.5 * rr * qr[IV] * qr[IV];
  if (NV == 5) ecinr += 0.5 * rr * qr[IW] * qr[IW];
  const double etotr = pr * entho + ecinr;
  // max(cfastl, cfastr) = sqrt(max of the radicands): the root is monotonic and correctly rounded (see fast_speed_sq)
  const double cfast = rg_sqrt(fmax(fmax(rg_div(g.gamma0 * pl, rg_recip(rl)), g.smallc * g.smallc),
                                    fmax(rg_div(g.gamma0 * pr, rg_recip(rr)), g.smallc * g.smallc)));
  const double SL = fmin(ul, ur) - cfast;
  const double SR = fmax(ul, ur) + cfast;
  const double rcl = rl * (ul - SL), rcr = rr * (SR - ur);
  const rg_recip_t inv_rc = rg_recip(rcr + rcl);
  const double ustar = rg_div(rcr * ur + rcl * ul + (pl - pr), inv_rc);
  const double ptotstar = rg_div(rcr * pl + rcl * pr + rcl * rcr * (ul - ur), inv_rc);
  const rg_recip_t inv_sl = rg_recip(SL - ustar), inv_sr = rg_recip(SR - ustar);
  const double rstarl = rg_div(rl * (SL - ul), inv_sl);
  const double etotstarl = rg_div((SL - ul) * etotl - pl * ul + ptotstar * ustar, inv_sl);
  const double rstarr = rg_div(rr * (SR - ur), inv_sr);
  const double etotstarr = rg_div((SR - ur) * etotr - pr * ur + ptotstar * ustar, inv_sr);
  double ro, uo, ptoto, etoto;
  if (SL > 0.0) { ro = rl; uo = ul; ptoto = pl; etoto = etotl; }
  else if (ustar > 0.0) { ro = rstarl; uo = ustar; ptoto = ptotstar; etoto = etotstarl; }
  else if (SR > 0.0) { ro = rstarr; uo = ustar; ptoto = ptotstar; etoto = etotstarr; }
  else { ro = rr; uo = ur; ptoto = pr; etoto = etotr; }
  flux[ID] = ro * uo;
  flux[IU] = ro * uo * uo + ptoto;
  flux[IP] = (etoto + ptoto) * uo;
  const bool upwind_left = flux[ID] > 0.0;
  flux[IV] = flux[ID] * (upwind_left ? ql[IV] : qr[IV]);
  if (NV == 5) flux[IW] = flux[ID] * (upwind_left ? ql[IW] : qr[IW]);
}

// riemann<NVAR> (riemann.h:388-401).  Other selections leave flux as passed in (callers zero it).
template <int NV>
RG_DEVFN void hydro_riemann(const DevParams& g, const double* ql, const double* qr, double* flux) {
  if (g.riemannSolver == 0) riemann_approx<NV>(g, ql, qr, flux);
  else if (g.riemannSolver == 1) riemann_hll<NV>(g, ql, qr, flux);
  else if (g.riemannSolver == 2) riemann_hllc<NV>(g, ql, qr, flux);
}

// ---------------------------------------------------------------------------------------------------------
// MHD: primitive variables, wave speeds
// ---------------------------------------------------------------------------------------------------------

// constoprim_mhd (constoprim.h:137-199).  bnx/bny/bnz: left-face field of the +1 neighbours (bnz = 0 in 2D,
// which makes Bz_cell = Bz/2 there: computePrimitives_MHD_2D, constoprim.h:405-409)
RG_DEVFN Prim8 mhd_prim(const DevParams& g, const double* u, double bnx, double bny, double bnz, double dt) {
  Prim8 q;
  q.r = fmax(u[ID], g.smallr);
  const rg_recip_t inv_r = rg_recip(q.r);
  q.u = rg_div(u[IU], inv_r);
  q.v = rg_div(u[IV], inv_r);
  q.w = rg_div(u[IW], inv_r);
  q.a = 0.5 * (u[IA] + bnx);
  q.b = 0.5 * (u[IB] + bny);
  q.c = 0.5 * (u[IC] + bnz);
  const double eken = 0.5 * (q.u * q.u + q.v * q.v + q.w * q.w);
  const double emag = 0.5 * (q.a * q.a + q.b * q.b + q.c * q.c);
  if (g.cIso > 0) {
    q.p = q.r * g.cIso * g.cIso;
  } else {
    const double eint = rg_div(u[IP] - emag, inv_r) - eken;
    q.p = fmax((g.gamma0 - 1.0) * q.r * eint, q.r * g.smallp);
  }
  if (g.Omega0 > 0) {  // Coriolis half-step predictor: both increments from the un-updated velocities
    const double dvx = 2.0 * g.Omega0 * q.v;
    const double dvy = -0.5 * g.Omega0 * q.u;
    q.u += dvx * dt * 0.5;
    q.v += dvy * dt * 0.5;
  }
  return q;
}

// find_speed_fast<IX> (mhd_utils.h:28-52), bn = the field component along the wanted direction
// inv_r = rg_recip(q.r): the three divisions by the density (and those of a second call for another direction
// of the same state) share one reciprocal
// fast_speed_sq: the radicand, cf = sqrt(fast_speed_sq).  The square root is monotonic and (exact arithmetic) correctly
// rounded, so the MAXIMUM of several fast speeds is the root of the maximum radicand, bit for bit: where the reference only
// consumes max(cf_1 .. cf_n) -- the wave-speed bounds of riemann_hlld / riemann_hll, mag_riemann2d_hlld / _hllf -- one root
// is taken instead of n (a NaN radicand is dropped by fmax exactly like the NaN root would be).
// ISO_P: the caller has set q.p = q.r cIso^2 if the gas is isothermal (the HLLD solvers do).  Contracted arithmetic then takes the
// sound speed squared as the constant it is, gamma0 cIso^2, instead of gamma0 (rho cIso cIso) / rho.
template <bool ISO_P = false>
RG_DEVFN double fast_speed_sq(const DevParams& g, const Prim8& q, double bn, const rg_recip_t& inv_r) {
  const double b2 = q.a * q.a + q.b * q.b + q.c * q.c;
#ifdef RG_ARITH_FAST
  const double c2 = (ISO_P && g.cIso > 0) ? g.gamma0 * (g.cIso * g.cIso) : rg_div(g.gamma0 * q.p, inv_r);
#else
  const double c2 = rg_div(g.gamma0 * q.p, inv_r);
#endif
  const double d2 = 0.5 * (rg_div(b2, inv_r) + c2);
  return d2 + rgpu::rg_sqrt_radicand(d2 * d2 - rg_div(c2 * bn * bn, inv_r));
}
RG_DEVFN double fast_speed(const DevParams& g, const Prim8& q, double bn, const rg_recip_t& inv_r) {
  return rg_sqrt_pos(fast_speed_sq(g, q, bn, inv_r));   // d2 > 0: p > 0
}
RG_DEVFN double fast_speed(const DevParams& g, const Prim8& q, double bn) { return fast_speed(g, q, bn, rg_recip(q.r)); }

// find_speed_info<NDIM> (mhd_utils.h:241-284): sum_d (cf_d + |v_d|)/delta_d is formed by the caller
RG_DEVFN void info_speeds(const DevParams& g, const Prim8& q, double& sx, double& sy, double& sz) {
  const double b2 = q.a * q.a + q.b * q.b + q.c * q.c;
  const rg_recip_t inv_r = rg_recip(q.r);
  const double c2 = rg_div(g.gamma0 * q.p, inv_r);
  const double d2 = 0.5 * (rg_div(b2, inv_r) + c2);
  sx = rg_sqrt(d2 + rg_sqrt(d2 * d2 - rg_div(c2 * q.a * q.a, inv_r))) + fabs(q.u);
  sy = rg_sqrt(d2 + rg_sqrt(d2 * d2 - rg_div(c2 * q.b * q.b, inv_r))) + fabs(q.v);
  sz = rg_sqrt(d2 + rg_sqrt(d2 * d2 - rg_div(c2 * q.c * q.c, inv_r))) + fabs(q.w);
}

// find_mhd_flux (mhd_utils.h:106-156): conservative vector and flux of a primitive state (normal frame)
RG_DEVFN void mhd_physical_flux(const DevParams& g, const Prim8& q, double* cv, double* ff) {
  const double p = (g.cIso > 0) ? q.r * g.cIso * g.cIso : q.p;
  const double entho = 1.0 / (g.gamma0 - 1.0);
  const double ecin = 0.5 * (q.u * q.u + q.v * q.v + q.w * q.w) * q.r;
  const double emag = 0.5 * (q.a * q.a + q.b * q.b + q.c * q.c);
  const double etot = p * entho + ecin + emag;
  const double ptot = p + emag;
  cv[ID] = q.r; cv[IP] = etot; cv[IU] = q.r * q.u; cv[IV] = q.r * q.v; cv[IW] = q.r * q.w;
  cv[IA] = q.a; cv[IB] = q.b; cv[IC] = q.c;
  ff[ID] = q.r * q.u;
  ff[IP] = (etot + ptot) * q.u - q.a * (q.a * q.u + q.b * q.v + q.c * q.w);
  ff[IU] = q.r * q.u * q.u - q.a * q.a + ptot;
  ff[IV] = q.r * q.u * q.v - q.a * q.b;
  ff[IW] = q.r * q.u * q.w - q.a * q.c;
  ff[IA] = 0.0;
  ff[IB] = q.b * q.u - q.a * q.v;
  ff[IC] = q.c * q.u - q.a * q.w;
}

// ---------------------------------------------------------------------------------------------------------
// MHD 1D Riemann solvers in the face-normal frame.  L and R are taken by reference: like the reference
// routines they are left with the averaged normal field (and the isothermal pressure), which the rotating
// path reuses for the shear correction of the y flux.
// ---------------------------------------------------------------------------------------------------------

// isothermal pressure rho cIso cIso (riemann_mhd.h:157-160, 1071-1076): the reference's two products; contracted arithmetic: one, by cIso^2
RG_DEVFN double iso_pressure(const DevParams& g, double r) {
#ifdef RG_ARITH_FAST
  return r * (g.cIso * g.cIso);
#else
  return r * g.cIso * g.cIso;
#endif
}

// riemann_hlld (riemann_mhd.h:140-342), Miyoshi & Kusano 2005
RG_DEVFN void mhd_hlld(const DevParams& g, Prim8& L, Prim8& R, double* flux) {
  const double entho = 1.0 / (g.gamma0 - 1.0);
  const double a = 0.5 * (L.a + R.a);
  const double sgnm = (a >= 0) ? 1.0 : -1.0;
  L.a = a;
  R.a = a;
  if (g.cIso > 0) {
    L.p = iso_pressure(g, L.r);
    R.p = iso_pressure(g, R.r);
  }
  const double rl = L.r, pl = L.p, ul = L.u, vl = L.v, wl = L.w, bl = L.b, cl = L.c;
  const double ecinl = 0.5 * (ul * ul + vl * vl + wl * wl) * rl;
  const double emagl = 0.5 * (a * a + bl * bl + cl * cl);
  const double etotl = pl * entho + ecinl + emagl;
  const double ptotl = pl + emagl;
  const double vdotbl = ul * a + vl * bl + wl * cl;
  const double rr = R.r, pr = R.p, ur = R.u, vr = R.v, wr = R.w, br = R.b, cr = R.c;
  const double ecinr = 0.5 * (ur * ur + vr * vr + wr * wr) * rr;
  const double emagr = 0.5 * (a * a + br * br + cr * cr);
  const double etotr = pr * entho + ecinr + emagr;
  const double ptotr = pr + emagr;
  const double vdotbr = ur * a + vr * br + wr * cr;
  rg_recip_t inv_rl, inv_rr;
  rg_recip2(L.r, R.r, inv_rl, inv_rr);
  const double cfast = rg_sqrt_pos(fmax(fast_speed_sq<true>(g, L, L.a, inv_rl), fast_speed_sq<true>(g, R, R.a, inv_rr)));   // = max(cfastl, cfastr)
  const double sl = fmin(ul, ur) - cfast;
  const double sr = fmax(ul, ur) + cfast;
  const double rcl = rl * (ul - sl), rcr = rr * (sr - ur);
  const rg_recip_t inv_rc = rg_recip(rcr + rcl);
  const double ustar = rg_div(rcr * ur + rcl * ul + (ptotl - ptotr), inv_rc);
  const double ptotstar = rg_div(rcr * ptotl + rcl * ptotr + rcl * rcr * (ul - ur), inv_rc);
  const double a2 = a * a;
  const rg_recip_t inv_a2 = rg_recip(a2);
  // left star state  (sl < ustar < sr: neither difference vanishes)
  rg_recip_t inv_sl, inv_sr;
  rg_recip2(sl - ustar, sr - ustar, inv_sl, inv_sr);
  const double rstarl = rg_div(rl * (sl - ul), inv_sl);
  const double estarl = rl * (sl - ul) * (sl - ustar) - a2;
  const double el = rl * (sl - ul) * (sl - ul) - a2;
  const rg_recip_t inv_el = rg_recip(estarl);
  const bool degl = (a2 > 0) && (fabs(rg_div(estarl, inv_a2) - 1.0) <= 1e-8);
  const double vstarl = degl ? vl : vl - rg_div(a * bl * (ustar - ul), inv_el);
  const double bstarl = degl ? bl : rg_div(bl * el, inv_el);
  const double wstarl = degl ? wl : wl - rg_div(a * cl * (ustar - ul), inv_el);
  const double cstarl = degl ? cl : rg_div(cl * el, inv_el);
  const double vdotbstarl = ustar * a + vstarl * bstarl + wstarl * cstarl;
  const double etotstarl = rg_div((sl - ul) * etotl - ptotl * ul + ptotstar * ustar + a * (vdotbl - vdotbstarl), inv_sl);
  const double sqrrstarl = rg_sqrt_pos(rstarl);
  // right star state
  const double rstarr = rg_div(rr * (sr - ur), inv_sr);
  const double estarr = rr * (sr - ur) * (sr - ustar) - a2;
  const double er = rr * (sr - ur) * (sr - ur) - a2;
  const rg_recip_t inv_er = rg_recip(estarr);
  const bool degr = (a2 > 0) && (fabs(rg_div(estarr, inv_a2) - 1.0) <= 1e-8);
  const double vstarr = degr ? vr : vr - rg_div(a * br * (ustar - ur), inv_er);
  const double bstarr = degr ? br : rg_div(br * er, inv_er);
  const double wstarr = degr ? wr : wr - rg_div(a * cr * (ustar - ur), inv_er);
  const double cstarr = degr ? cr : rg_div(cr * er, inv_er);
  const double vdotbstarr = ustar * a + vstarr * bstarr + wstarr * cstarr;
  const double etotstarr = rg_div((sr - ur) * etotr - ptotr * ur + ptotstar * ustar + a * (vdotbr - vdotbstarr), inv_sr);
  const double sqrrstarr = rg_sqrt_pos(rstarr);
  rg_recip_t inv_ql, inv_qr;
  rg_recip2(sqrrstarl, sqrrstarr, inv_ql, inv_qr);
  const double sal = ustar - rg_div(fabs(a), inv_ql);
  const double sar = ustar + rg_div(fabs(a), inv_qr);
  // double star state
  const rg_recip_t inv_sq = rg_recip(sqrrstarl + sqrrstarr);
  const double vstarstar = rg_div(sqrrstarl * vstarl + sqrrstarr * vstarr + sgnm * (bstarr - bstarl), inv_sq);
  const double wstarstar = rg_div(sqrrstarl * wstarl + sqrrstarr * wstarr + sgnm * (cstarr - cstarl), inv_sq);
  const double bstarstar = rg_div(sqrrstarl * bstarr + sqrrstarr * bstarl + sgnm * sqrrstarl * sqrrstarr * (vstarr - vstarl), inv_sq);
  const double cstarstar = rg_div(sqrrstarl * cstarr + sqrrstarr * cstarl + sgnm * sqrrstarl * sqrrstarr * (wstarr - wstarl), inv_sq);
  const double vdotbstarstar = ustar * a + vstarstar * bstarstar + wstarstar * cstarstar;
  const double etotstarstarl = etotstarl - sgnm * sqrrstarl * (vdotbstarl - vdotbstarstar);
  const double etotstarstarr = etotstarr + sgnm * sqrrstarr * (vdotbstarr - vdotbstarstar);
  // sample at x/t = 0
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

// riemann_hll (riemann_mhd.h:42-71)
RG_DEVFN void mhd_hll(const DevParams& g, Prim8& L, Prim8& R, double* flux) {
  const double bx_mean = 0.5 * (L.a + R.a);
  L.a = bx_mean;
  R.a = bx_mean;
  double uL[8], fL[8], uR[8], fR[8];
  mhd_physical_flux(g, L, uL, fL);
  mhd_physical_flux(g, R, uR, fR);
  const double cf = rg_sqrt_pos(fmax(fast_speed_sq(g, L, L.a, rg_recip(L.r)), fast_speed_sq(g, R, R.a, rg_recip(R.r))));   // = max(cfl, cfr)
  const double sl = fmin(fmin(L.u, R.u) - cf, 0.0);
  const double sr = fmax(fmax(L.u, R.u) + cf, 0.0);
#pragma unroll
  for (int n = 0; n < 8; ++n) flux[n] = (sr * fL[n] - sl * fR[n] + sr * sl * (uR[n] - uL[n])) / (sr - sl);
}

// riemann_llf (riemann_mhd.h:87-118) -- including its mean of the STATES (not of the fluxes)
RG_DEVFN void mhd_llf(const DevParams& g, Prim8& L, Prim8& R, double* flux, double zero_flux = 1.0) {
  const double bx_mean = 0.5 * (L.a + R.a);
  L.a = bx_mean;
  R.a = bx_mean;
  double uL[8], fL[8], uR[8], fR[8];
  mhd_physical_flux(g, L, uL, fL);
  mhd_physical_flux(g, R, uR, fR);
  const double ql[8] = {L.r, L.p, L.u, L.v, L.w, L.a, L.b, L.c};
  const double qr[8] = {R.r, R.p, R.u, R.v, R.w, R.a, R.b, R.c};
  const double vel_info = fmax(fast_speed(g, L, L.a) + fabs(L.u), fast_speed(g, R, R.a) + fabs(R.u));
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    flux[n] = (ql[n] + qr[n]) / 2 * zero_flux;
    flux[n] -= vel_info * (uR[n] - uL[n]) / 2;
  }
}

// riemann_mhd (riemann_mhd.h:355-368): approx / hllc selections compute nothing (flux stays 0)
RG_DEVFN void mhd_riemann(const DevParams& g, Prim8& L, Prim8& R, double* flux) {
  if (g.riemannSolver == 3) mhd_hlld(g, L, R, flux);
  else if (g.riemannSolver == 1) mhd_hll(g, L, R, flux);
  else if (g.riemannSolver == 4) mhd_llf(g, L, R, flux);
}

// ---------------------------------------------------------------------------------------------------------
// 2D magnetic Riemann problem at a cell edge (the EMF of constrained transport)
// ---------------------------------------------------------------------------------------------------------
// The reference's FMAX / FMIN chains are "a1 > ret ? a1 : ret" selections.  On operands without NaN they differ from the
// IEEE maximum / minimum (v_max_f64 / v_min_f64: one instruction instead of a compare and two selects) only in the sign of a
// zero result.  The max_of4 / min_of4 / pos_max chains below are therefore used where that sign cannot matter: all operands
// non-negative magnitudes (speeds: square roots, |b| / sqrt(rho), smallc -- never -0), or a velocity extremum that is at once
// added to / subtracted from a strictly positive fast speed.  sel_* keep the selection semantics for the rest.
RG_DEVFN double sel_max(double a0, double a1) { return (a1 > a0) ? a1 : a0; }
RG_DEVFN double sel_min(double a0, double a1) { return (a1 < a0) ? a1 : a0; }
RG_DEVFN double sel_max_of4(double a0, double a1, double a2, double a3) { return sel_max(sel_max(sel_max(a0, a1), a2), a3); }
RG_DEVFN double sel_min_of4(double a0, double a1, double a2, double a3) { return sel_min(sel_min(sel_min(a0, a1), a2), a3); }
RG_DEVFN double pos_max(double a0, double a1) { return fmax(a0, a1); }
RG_DEVFN double max_of4(double a0, double a1, double a2, double a3) { return fmax(fmax(fmax(a0, a1), a2), a3); }
RG_DEVFN double min_of4(double a0, double a1, double a2, double a3) { return fmin(fmin(fmin(a0, a1), a2), a3); }

// ---- Alfven speeds of mag_riemann2d_hlld (riemann_mhd.h:727-738): which candidate wins, decided without taking roots ----
// calfvenX = max of four |b_i| / sqrt(rho_i) and smallc, each candidate the correctly rounded quotient by the correctly rounded
// root.  The four come in two pairs, one per state: (|b|, rho_x) and its "star" partner (|b| t, rho_x t) with t = dv / dS, both
// built by fl(fl(. * dv) / dS).  In real arithmetic partner / plain = sqrt(t); every rounding on the way (two per star quantity,
// two per root-and-quotient of either candidate: <= 18 of 2^-53 on the squares) moves that ratio by < 2^-48, so |dv - dS| > 2^-40 |dS| settles the order of the
// two COMPUTED candidates (alfven_pick).  The two survivors of a group are compared through b1^2 rho2 against b2^2 rho1 (four
// roundings against the eight of the computed candidates' squares: < 2^-49): a margin of 2^-45 settles it (alfven_duel).  Anything
// closer -- a uniform state, an exact symmetry -- is "unsure" and the caller runs the reference's own sequence for the whole
// wave.  Products below 2^-900 (fields below ~1e-117 at densities within 2^+-60: underflow makes the products imprecise) are not
// compared at all: such candidates lose against smallc (>= 1e-100 is checked by the caller), whichever is picked.  The fp64
// maximum is monotonic, so evaluating only the winner returns the bits the reference's FMAX5 chain returns.
#ifndef RG_ALFVEN_SELECT
#ifdef RG_ARITH_FAST
#define RG_ALFVEN_SELECT 0   // contracted arithmetic: a root costs five instructions there, the selection would not pay
#else
#define RG_ALFVEN_SELECT 1
#endif
#endif
struct AlfvenPair { double b, r; };
RG_DEVFN AlfvenPair alfven_pick(double b, double rho, double bstar, double rhostar, double dv, double dS, bool& unsure) {
  const bool star = fabs(dv) > fabs(dS);
  unsure = unsure || !(fabs(dv - dS) > 0x1p-40 * fabs(dS));
  AlfvenPair w;
  w.b = star ? bstar : b;
  w.r = star ? rhostar : rho;
  return w;
}
RG_DEVFN AlfvenPair alfven_duel(const AlfvenPair& c1, const AlfvenPair& c2, bool& unsure) {
  const double p1 = (c1.b * c1.b) * c2.r, p2 = (c2.b * c2.b) * c1.r;
  const bool first = p1 >= p2;
  const double hi = fmax(p1, p2), lo = fmin(p1, p2);
  unsure = unsure || (!(lo * (1.0 + 0x1p-45) < hi) && hi > 0x1p-900);
  return first ? c1 : c2;
}

// mag_riemann2d_hlld (riemann_mhd.h:616-821).  States are in the edge frame (u,v = the two in-plane
// velocities, a,b = the two in-plane field components); E?? = u*b - v*a of each state.
// FORCE_REF / route: the self-test of the Alfven selection (rgpu_selftest_alfven) runs every sample twice -- the selection and, FORCE_REF,
// the reference's own sequence -- and learns through *route (1 = the reference's sequence ran) which way the lane's wave went.
template <bool FORCE_REF = false>
RG_DEVFN double mag_hlld_2d(const DevParams& g, const Prim8& LL, const Prim8& RL, const Prim8& LR, const Prim8& RR,
                            double ELL, double ERL, double ELR, double ERR, int* route = 0) {
  // 66 divisions by 24 distinct denominators: every denominator gets one shared reciprocal (rg_recip), see
  // rg_backend.h; numerators and operand order are the reference's
  rg_recip_t iLLr, iLRr, iRLr, iRRr;
  rg_recip4(LL.r, LR.r, RL.r, RR.r, iLLr, iLRr, iRLr, iRRr);
  // the eight fast speeds are only consumed through their maxima per direction: two roots instead of eight (fast_speed_sq)
  // (edge_emf has set the isothermal pressures: fast_speed_sq<true>)
  const double cxmax = rg_sqrt_pos(max_of4(fast_speed_sq<true>(g, LL, LL.a, iLLr), fast_speed_sq<true>(g, LR, LR.a, iLRr),
                                           fast_speed_sq<true>(g, RL, RL.a, iRLr), fast_speed_sq<true>(g, RR, RR.a, iRRr)));
  const double cymax = rg_sqrt_pos(max_of4(fast_speed_sq<true>(g, LL, LL.b, iLLr), fast_speed_sq<true>(g, LR, LR.b, iLRr),
                                           fast_speed_sq<true>(g, RL, RL.b, iRLr), fast_speed_sq<true>(g, RR, RR.b, iRRr)));
  const double SL = min_of4(LL.u, LR.u, RL.u, RR.u) - cxmax;
  const double SR = max_of4(LL.u, LR.u, RL.u, RR.u) + cxmax;
  const double SB = min_of4(LL.v, LR.v, RL.v, RR.v) - cymax;
  const double ST = max_of4(LL.v, LR.v, RL.v, RR.v) + cymax;
  const double PtotLL = LL.p + 0.5 * (LL.a * LL.a + LL.b * LL.b + LL.c * LL.c);
  const double PtotLR = LR.p + 0.5 * (LR.a * LR.a + LR.b * LR.b + LR.c * LR.c);
  const double PtotRL = RL.p + 0.5 * (RL.a * RL.a + RL.b * RL.b + RL.c * RL.c);
  const double PtotRR = RR.p + 0.5 * (RR.a * RR.a + RR.b * RR.b + RR.c * RR.c);
  const double rcLLx = LL.r * (LL.u - SL), rcRLx = RL.r * (SR - RL.u);
  const double rcLRx = LR.r * (LR.u - SL), rcRRx = RR.r * (SR - RR.u);
  const double rcLLy = LL.r * (LL.v - SB), rcLRy = LR.r * (ST - LR.v);
  const double rcRLy = RL.r * (RL.v - SB), rcRRy = RR.r * (ST - RR.v);
  rg_recip_t irx, iry;   // (sums of positive rho (u - S) terms)
  rg_recip2(rcLLx + rcLRx + rcRLx + rcRRx, rcLLy + rcLRy + rcRLy + rcRRy, irx, iry);
  const double ustar = rg_div(rcLLx * LL.u + rcLRx * LR.u + rcRLx * RL.u + rcRRx * RR.u + (PtotLL - PtotRL + PtotLR - PtotRR), irx);
  const double vstar = rg_div(rcLLy * LL.v + rcLRy * LR.v + rcRLy * RL.v + rcRRy * RR.v + (PtotLL - PtotLR + PtotRL - PtotRR), iry);
  rg_recip_t iSL, iSR, iSB, iST;   // (SL < ustar < SR, SB < vstar < ST)
  rg_recip4(SL - ustar, SR - ustar, SB - vstar, ST - vstar, iSL, iSR, iSB, iST);
  // per-state star quantities.  rstar = r*(S-u)/(S-ustar) is needed twice in the reference (alone and inside
  // the product with the y ratio); the identical sub-expression gives the identical value.
#ifdef RG_ARITH_FAST
  // contracted arithmetic: the ratios (S - u) / (S - ustar), (S - v) / (S - vstar) of a state once, then one product per quantity
  const double tLLx = (SL - LL.u) * iSL.r, tLLy = (SB - LL.v) * iSB.r;
  const double rstarLLx = LL.r * tLLx, BstarLL = LL.b * tLLx, rstarLLy = LL.r * tLLy, AstarLL = LL.a * tLLy, rstarLL = rstarLLx * tLLy;
  const double tLRx = (SL - LR.u) * iSL.r, tLRy = (ST - LR.v) * iST.r;
  const double rstarLRx = LR.r * tLRx, BstarLR = LR.b * tLRx, rstarLRy = LR.r * tLRy, AstarLR = LR.a * tLRy, rstarLR = rstarLRx * tLRy;
  const double tRLx = (SR - RL.u) * iSR.r, tRLy = (SB - RL.v) * iSB.r;
  const double rstarRLx = RL.r * tRLx, BstarRL = RL.b * tRLx, rstarRLy = RL.r * tRLy, AstarRL = RL.a * tRLy, rstarRL = rstarRLx * tRLy;
  const double tRRx = (SR - RR.u) * iSR.r, tRRy = (ST - RR.v) * iST.r;
  const double rstarRRx = RR.r * tRRx, BstarRR = RR.b * tRRx, rstarRRy = RR.r * tRRy, AstarRR = RR.a * tRRy, rstarRR = rstarRRx * tRRy;
#else
  const double rstarLLx = rg_div(LL.r * (SL - LL.u), iSL);
  const double BstarLL = rg_div(LL.b * (SL - LL.u), iSL);
  const double rstarLLy = rg_div(LL.r * (SB - LL.v), iSB);
  const double AstarLL = rg_div(LL.a * (SB - LL.v), iSB);
  const double rstarLL = rg_div(rstarLLx * (SB - LL.v), iSB);

  const double rstarLRx = rg_div(LR.r * (SL - LR.u), iSL);
  const double BstarLR = rg_div(LR.b * (SL - LR.u), iSL);
  const double rstarLRy = rg_div(LR.r * (ST - LR.v), iST);
  const double AstarLR = rg_div(LR.a * (ST - LR.v), iST);
  const double rstarLR = rg_div(rstarLRx * (ST - LR.v), iST);

  const double rstarRLx = rg_div(RL.r * (SR - RL.u), iSR);
  const double BstarRL = rg_div(RL.b * (SR - RL.u), iSR);
  const double rstarRLy = rg_div(RL.r * (SB - RL.v), iSB);
  const double AstarRL = rg_div(RL.a * (SB - RL.v), iSB);
  const double rstarRL = rg_div(rstarRLx * (SB - RL.v), iSB);

  const double rstarRRx = rg_div(RR.r * (SR - RR.u), iSR);
  const double BstarRR = rg_div(RR.b * (SR - RR.u), iSR);
  const double rstarRRy = rg_div(RR.r * (ST - RR.v), iST);
  const double AstarRR = rg_div(RR.a * (ST - RR.v), iST);
  const double rstarRR = rg_div(rstarRRx * (ST - RR.v), iST);

#endif

  // FMAX5 chains (riemann_mhd.h:401-411, 727-738): "a1 > ret ? a1 : ret" selections in argument order
  double calfvenL, calfvenR, calfvenB, calfvenT;
#if RG_ALFVEN_SELECT
  // Sixteen |b| / sqrt(rho) candidates, twelve distinct roots, four maxima consumed: pick each group's winner BEFORE any root is
  // taken and evaluate only the winner with the reference's two operations (alfven_pick / alfven_duel above).  A lane whose
  // ordering is not certain at fp64 round-off sends its wave down the reference's own sequence.
  AlfvenPair wL, wR, wB, wT;
  bool unsure = g.smallc < 1e-100;   // (candidates below ~1e-117 are assumed to lose against smallc: see alfven_duel)
  {
    const AlfvenPair lr = alfven_pick(LR.a, rstarLRx, AstarLR, rstarLR, ST - LR.v, iST.d, unsure);
    const AlfvenPair ll = alfven_pick(LL.a, rstarLLx, AstarLL, rstarLL, SB - LL.v, iSB.d, unsure);
    wL = alfven_duel(lr, ll, unsure);
    const AlfvenPair rr = alfven_pick(RR.a, rstarRRx, AstarRR, rstarRR, ST - RR.v, iST.d, unsure);
    const AlfvenPair rl = alfven_pick(RL.a, rstarRLx, AstarRL, rstarRL, SB - RL.v, iSB.d, unsure);
    wR = alfven_duel(rr, rl, unsure);
    const AlfvenPair bll = alfven_pick(LL.b, rstarLLy, BstarLL, rstarLL, SL - LL.u, iSL.d, unsure);
    const AlfvenPair brl = alfven_pick(RL.b, rstarRLy, BstarRL, rstarRL, SR - RL.u, iSR.d, unsure);
    wB = alfven_duel(bll, brl, unsure);
    const AlfvenPair blr = alfven_pick(LR.b, rstarLRy, BstarLR, rstarLR, SL - LR.u, iSL.d, unsure);
    const AlfvenPair brr = alfven_pick(RR.b, rstarRRy, BstarRR, rstarRR, SR - RR.u, iSR.d, unsure);
    wT = alfven_duel(blr, brr, unsure);
  }
  const bool reference_sequence = FORCE_REF || rgpu::rg_wave_any(unsure);
  if (route) *route = reference_sequence ? 1 : 0;
  if (!reference_sequence) {
    calfvenL = pos_max(rg_div(fabs(wL.b), rg_recip_sqrt_pos(wL.r)), g.smallc);
    calfvenR = pos_max(rg_div(fabs(wR.b), rg_recip_sqrt_pos(wR.r)), g.smallc);
    calfvenB = pos_max(rg_div(fabs(wB.b), rg_recip_sqrt_pos(wB.r)), g.smallc);
    calfvenT = pos_max(rg_div(fabs(wT.b), rg_recip_sqrt_pos(wT.r)), g.smallc);
  } else
#else
  if (route) *route = 1;
#endif
  {
  const rg_recip_t iqLL = rg_recip_sqrt_pos(rstarLL), iqLR = rg_recip_sqrt_pos(rstarLR);
  const rg_recip_t iqRL = rg_recip_sqrt_pos(rstarRL), iqRR = rg_recip_sqrt_pos(rstarRR);
  calfvenL = pos_max(pos_max(pos_max(pos_max(rg_div(fabs(LR.a), rg_recip_sqrt_pos(rstarLRx)), rg_div(fabs(AstarLR), iqLR)),
                                                  rg_div(fabs(LL.a), rg_recip_sqrt_pos(rstarLLx))), rg_div(fabs(AstarLL), iqLL)), g.smallc);
  calfvenR = pos_max(pos_max(pos_max(pos_max(rg_div(fabs(RR.a), rg_recip_sqrt_pos(rstarRRx)), rg_div(fabs(AstarRR), iqRR)),
                                                  rg_div(fabs(RL.a), rg_recip_sqrt_pos(rstarRLx))), rg_div(fabs(AstarRL), iqRL)), g.smallc);
  calfvenB = pos_max(pos_max(pos_max(pos_max(rg_div(fabs(LL.b), rg_recip_sqrt_pos(rstarLLy)), rg_div(fabs(BstarLL), iqLL)),
                                                  rg_div(fabs(RL.b), rg_recip_sqrt_pos(rstarRLy))), rg_div(fabs(BstarRL), iqRL)), g.smallc);
  calfvenT = pos_max(pos_max(pos_max(pos_max(rg_div(fabs(LR.b), rg_recip_sqrt_pos(rstarLRy)), rg_div(fabs(BstarLR), iqLR)),
                                                  rg_div(fabs(RR.b), rg_recip_sqrt_pos(rstarRRy))), rg_div(fabs(BstarRR), iqRR)), g.smallc);
  }
  const double SAL = fmin(ustar - calfvenL, 0.0);
  const double SAR = fmax(ustar + calfvenR, 0.0);
  const double SAB = fmin(vstar - calfvenB, 0.0);
  const double SAT = fmax(vstar + calfvenT, 0.0);
  rg_recip_t iSA, iSAy;   // (the Alfven speeds are at least smallc: both differences are positive)
  rg_recip2(SAR - SAL, SAT - SAB, iSA, iSAy);

  // Region selection by sign bits.  The reference evaluates all five candidate values and adds each multiplied by its
  // 0 / 1 integer mask (riemann_mhd.h:759-787).  A term with mask 0 adds (0 * finite value) = +-0, which leaves the sum
  // unchanged, so only the terms whose mask is 1 are evaluated here -- in a wave whose lanes agree on the region (the
  // usual case) the other four are never computed.  (Only difference: a non-finite value in an unselected candidate
  // would turn the reference's sum into NaN.)
  const int SB_pos = signbit(SB) ? 0 : 1, SB_neg = 1 - SB_pos;
  const int ST_pos = signbit(ST) ? 0 : 1, ST_neg = 1 - ST_pos;
  const int SL_pos = signbit(SL) ? 0 : 1, SL_neg = 1 - SL_pos;
  const int SR_pos = signbit(SR) ? 0 : 1, SR_neg = 1 - SR_pos;
  double E = 0, tmpE;
  if (SB_neg * ST_pos * SL_neg * SR_pos) {
    const double AstarT = rg_div(SAR * AstarRR - SAL * AstarLR, iSA);
    const double AstarB = rg_div(SAR * AstarRL - SAL * AstarLL, iSA);
    const double BstarR = rg_div(SAT * BstarRR - SAB * BstarRL, iSAy);
    const double BstarL = rg_div(SAT * BstarLR - SAB * BstarLL, iSAy);
    const double EstarLL = ustar * BstarLL - vstar * AstarLL, EstarLR = ustar * BstarLR - vstar * AstarLR;
    const double EstarRL = ustar * BstarRL - vstar * AstarRL, EstarRR = ustar * BstarRR - vstar * AstarRR;
    tmpE = rg_div(rg_div(SAL * SAB * EstarRR - SAL * SAT * EstarRL - SAR * SAB * EstarLR + SAR * SAT * EstarLL, iSA), iSAy) -
           rg_div(SAT * SAB, iSAy) * (AstarT - AstarB) + rg_div(SAR * SAL, iSA) * (BstarR - BstarL);
    E += (double)(SB_neg * ST_pos * SL_neg * SR_pos) * tmpE;
  }
  if (SB_pos) {
    const double EstarLLx = ustar * BstarLL - LL.v * LL.a, EstarRLx = ustar * BstarRL - RL.v * RL.a;
    tmpE = rg_div(SAR * EstarLLx - SAL * EstarRLx + SAR * SAL * (RL.b - LL.b), iSA);
    tmpE = (double)SL_pos * ELL + (double)(SL_neg * SR_neg) * ERL + (double)(SL_neg * SR_pos) * tmpE;
    E += (double)SB_pos * tmpE;
  }
  if (SB_neg * ST_neg) {
    const double EstarLRx = ustar * BstarLR - LR.v * LR.a, EstarRRx = ustar * BstarRR - RR.v * RR.a;
    tmpE = rg_div(SAR * EstarLRx - SAL * EstarRRx + SAR * SAL * (RR.b - LR.b), iSA);
    tmpE = (double)SL_pos * ELR + (double)(SL_neg * SR_neg) * ERR + (double)(SL_neg * SR_pos) * tmpE;
    E += (double)(SB_neg * ST_neg) * tmpE;
  }
  if (SB_neg * ST_pos * SL_pos) {
    const double EstarLLy = LL.u * LL.b - vstar * AstarLL, EstarLRy = LR.u * LR.b - vstar * AstarLR;
    tmpE = rg_div(SAT * EstarLLy - SAB * EstarLRy - SAT * SAB * (LR.a - LL.a), iSAy);
    E += (double)(SB_neg * ST_pos * SL_pos) * tmpE;
  }
  if (SB_neg * ST_pos * SL_neg * SR_neg) {
    const double EstarRLy = RL.u * RL.b - vstar * AstarRL, EstarRRy = RR.u * RR.b - vstar * AstarRR;
    tmpE = rg_div(SAT * EstarRLy - SAB * EstarRRy - SAT * SAB * (RR.a - RL.a), iSAy);
    E += (double)(SB_neg * ST_pos * SL_neg * SR_neg) * tmpE;
  }
  return E;
}

// the HLL average shared by mag_riemann2d_hlla / hllf (riemann_mhd.h:449-453, 497-501)
RG_DEVFN double mag_hll_average(const Prim8& LL, const Prim8& RR, double ELL, double ERL, double ELR, double ERR,
                                double SL, double SR, double SB, double ST) {
  const rg_recip_t iS = rg_recip(SR - SL), iT = rg_recip(ST - SB);
  return rg_div(rg_div(SL * SB * ERR - SL * ST * ERL - SR * SB * ELR + SR * ST * ELL, iS), iT) -
         rg_div(ST * SB, iT) * (RR.a - LL.a) + rg_div(SR * SL, iS) * (RR.b - LL.b);
}

// mag_riemann2d_hlla (riemann_mhd.h:417-457): Alfven speeds sqrt(b*b/rho) (mhd_utils.h:82-88), floor smallc
RG_DEVFN double mag_hlla_2d(const DevParams& g, const Prim8& LL, const Prim8& RL, const Prim8& LR, const Prim8& RR,
                            double ELL, double ERL, double ELR, double ERR) {
  const rg_recip_t iLL = rg_recip(LL.r), iLR = rg_recip(LR.r), iRL = rg_recip(RL.r), iRR = rg_recip(RR.r);
  const double cMaxx = pos_max(max_of4(rg_sqrt(rg_div(LL.a * LL.a, iLL)), rg_sqrt(rg_div(LR.a * LR.a, iLR)),
                                       rg_sqrt(rg_div(RL.a * RL.a, iRL)), rg_sqrt(rg_div(RR.a * RR.a, iRR))), g.smallc);
  const double cMaxy = pos_max(max_of4(rg_sqrt(rg_div(LL.b * LL.b, iLL)), rg_sqrt(rg_div(LR.b * LR.b, iLR)),
                                       rg_sqrt(rg_div(RL.b * RL.b, iRL)), rg_sqrt(rg_div(RR.b * RR.b, iRR))), g.smallc);
  const double SL = fmin(sel_min_of4(LL.u, LR.u, RL.u, RR.u) - cMaxx, 0.0);
  const double SR = fmax(sel_max_of4(LL.u, LR.u, RL.u, RR.u) + cMaxx, 0.0);
  const double SB = fmin(sel_min_of4(LL.v, LR.v, RL.v, RR.v) - cMaxy, 0.0);
  const double ST = fmax(sel_max_of4(LL.v, LR.v, RL.v, RR.v) + cMaxy, 0.0);
  return mag_hll_average(LL, RR, ELL, ERL, ELR, ERR, SL, SR, SB, ST);
}

// mag_riemann2d_hllf (riemann_mhd.h:463-505): fast magnetosonic speeds
RG_DEVFN double mag_hllf_2d(const DevParams& g, const Prim8& LL, const Prim8& RL, const Prim8& LR, const Prim8& RR,
                            double ELL, double ERL, double ELR, double ERR) {
  const rg_recip_t iLL = rg_recip(LL.r), iLR = rg_recip(LR.r), iRL = rg_recip(RL.r), iRR = rg_recip(RR.r);
  const double cMaxx = rg_sqrt_pos(max_of4(fast_speed_sq(g, LL, LL.a, iLL), fast_speed_sq(g, LR, LR.a, iLR), fast_speed_sq(g, RL, RL.a, iRL), fast_speed_sq(g, RR, RR.a, iRR)));
  const double cMaxy = rg_sqrt_pos(max_of4(fast_speed_sq(g, LL, LL.b, iLL), fast_speed_sq(g, LR, LR.b, iLR), fast_speed_sq(g, RL, RL.b, iRL), fast_speed_sq(g, RR, RR.b, iRR)));
  const double SL = fmin(min_of4(LL.u, LR.u, RL.u, RR.u) - cMaxx, 0.0);
  const double SR = fmax(max_of4(LL.u, LR.u, RL.u, RR.u) + cMaxx, 0.0);
  const double SB = fmin(min_of4(LL.v, LR.v, RL.v, RR.v) - cMaxy, 0.0);
  const double ST = fmax(max_of4(LL.v, LR.v, RL.v, RR.v) + cMaxy, 0.0);
  return mag_hll_average(LL, RR, ELL, ERL, ELR, ERR, SL, SR, SB, ST);
}

// mag_riemann2d_llf (riemann_mhd.h:517-609): mean of the four corner E + the IB component of two 1D LLF fluxes
// (zero_flux = 0) between face-averaged states, the second one with u<->v and a<->b swapped
RG_DEVFN double mag_llf_2d(const DevParams& g, const Prim8& LL, const Prim8& RL, const Prim8& LR, const Prim8& RR,
                           double ELL, double ERL, double ELR, double ERR) {
  double E = (ELL + ERL + ELR + ERR) / 4;
  double fx[8], fy[8];
  Prim8 l, r;
  l.r = (LL.r + LR.r) / 2; r.r = (RR.r + RL.r) / 2;
  l.p = (LL.p + LR.p) / 2; r.p = (RR.p + RL.p) / 2;
  l.u = (LL.u + LR.u) / 2; r.u = (RR.u + RL.u) / 2;
  l.v = (LL.v + LR.v) / 2; r.v = (RR.v + RL.v) / 2;
  l.w = (LL.w + LR.w) / 2; r.w = (RR.w + RL.w) / 2;
  l.a = (LL.a + LR.a) / 2; r.a = (RR.a + RL.a) / 2;
  l.b = (LL.b + LR.b) / 2; r.b = (RR.b + RL.b) / 2;
  l.c = (LL.c + LR.c) / 2; r.c = (RR.c + RL.c) / 2;
  mhd_llf(g, l, r, fx, 0.0);
  l.r = (LL.r + RL.r) / 2; r.r = (RR.r + LR.r) / 2;
  l.p = (LL.p + RL.p) / 2; r.p = (RR.p + LR.p) / 2;
  l.u = (LL.v + RL.v) / 2; r.u = (RR.v + LR.v) / 2;
  l.v = (LL.u + RL.u) / 2; r.v = (RR.u + LR.u) / 2;
  l.w = (LL.w + RL.w) / 2; r.w = (RR.w + LR.w) / 2;
  l.a = (LL.b + RL.b) / 2; r.a = (RR.b + LR.b) / 2;
  l.b = (LL.a + RL.a) / 2; r.b = (RR.a + LR.a) / 2;
  l.c = (LL.c + RL.c) / 2; r.c = (RR.c + LR.c) / 2;
  mhd_llf(g, l, r, fy, 0.0);
  E += (fx[IB] - fy[IB]);
  return E;
}

// compute_emf<dir> (riemann_mhd.h:1054-1193) on four edge states ALREADY in the edge frame
// (u,v,w = velocity along t1,t2,e ; a,b,c = field along t1,t2,e) and in the reference's slot order
// sRT, sRB, sLT, sLB.  EDIR: 0 = EMFX, 1 = EMFY, 2 = EMFZ.
template <int EDIR>
RG_DEVFN double edge_emf(const DevParams& g, const Prim8& sRT, const Prim8& sRB, const Prim8& sLT, const Prim8& sLB,
                         double xPos) {
  Prim8 LL = sRT, RL = sLT, LR = sRB, RR = sLB;
  if (g.cIso > 0) {
    LL.p = iso_pressure(g, LL.r);
    RL.p = iso_pressure(g, RL.r);
    LR.p = iso_pressure(g, LR.r);
    RR.p = iso_pressure(g, RR.r);
  }
  // enforce continuity of the two in-plane field components across the faces meeting at the edge
  const double aT = 0.5 * (sRT.a + sLT.a), aB = 0.5 * (sRB.a + sLB.a);
  const double bR = 0.5 * (sRT.b + sRB.b), bL = 0.5 * (sLT.b + sLB.b);
  LL.a = aT; RL.a = aT; LR.a = aB; RR.a = aB;
  LL.b = bR; RL.b = bL; LR.b = bR; RR.b = bL;
  const double ELL = LL.u * LL.b - LL.v * LL.a;
  const double ERL = RL.u * RL.b - RL.v * RL.a;
  const double ELR = LR.u * LR.b - LR.v * LR.a;
  const double ERR = RR.u * RR.b - RR.v * RR.a;
  double emf = 0;
  // MagneticRiemannSolverType (constants.h:149-156): 0 hlld, 1 hllf, 2 hlla, 4 llf
  if (g.magRiemannSolver == 0) emf = mag_hlld_2d(g, LL, RL, LR, RR, ELL, ERL, ELR, ERR);
  else if (g.magRiemannSolver == 2) emf = mag_hlla_2d(g, LL, RL, LR, RR, ELL, ERL, ELR, ERR);
  else if (g.magRiemannSolver == 1) emf = mag_hllf_2d(g, LL, RL, LR, RR, ELL, ERL, ELR, ERR);
  else if (g.magRiemannSolver == 4) emf = mag_llf_2d(g, LL, RL, LR, RR, ELL, ERL, ELR, ERR);
  if (g.Omega0 > 0) {  // upwinded shear advection of the field in the shearing box
    if (EDIR == 0) {
      const double shear = -1.5 * g.Omega0 * xPos;
      emf += shear * ((shear > 0) ? LL.b : RR.b);
    }
    if (EDIR == 2) {
      const double shear = -1.5 * g.Omega0 * (xPos - g.dx / 2);
      emf -= shear * ((shear > 0) ? LL.a : RR.a);
    }
  }
  return emf;
}

}  // namespace rgpu_dev
