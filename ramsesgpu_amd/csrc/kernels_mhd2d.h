// kernels_mhd2d.h -- per-cell bodies of the 2D MHD unsplit step ("implementationVersion 1"), and of the 2D branch of
// the rotating-frame step (Omega0 > 0, godunov_unsplit_rotating_cpu, MHDRunGodunov.cpp:2089-2434).
//   mhd_prim_cell (kernels_mhd3d.h)  U     -> Q    (8)   MHDRunGodunov.cpp:479-517 (Bz_cell = Bz/2 in 2D)
//   mhd_trace2d_cell                 U,Q   -> T2   (26)  mhd_godunov_unsplit_cpu_v1.cpp:43-94, trace_mhd.h:38-339
//   mhd_flux2d_cell                  T2    -> F2   (13)  ..._cpu_v1.cpp:99-222  (2 HLLD + emfZ)
//   mhd_update2d_cell                Uold,F2 -> Unew     ..._cpu_v1.cpp:171-197, 234-240
// T2 is the compact traced state (see kernels_mhd3d.h): advanced cell state, advanced low-face Bx/By, the
// x and y half slopes and the two transverse face half slopes.  Ez is recomputed inline from Q and the faces.
#pragma once
#include "kernels_mhd3d.h"

namespace rgpu_dev {

enum {
  T2_R = 0, T2_P, T2_U, T2_V, T2_W, T2_A, T2_B, T2_C,
  T2_AL, T2_BL,
  T2_DX,                  // 10..16: x half slopes of r,p,u,v,w,B,C
  T2_DY = T2_DX + 7,      // 17..23: y half slopes of r,p,u,v,w,A,C
  T2_DALY = T2_DY + 7, T2_DBLX,
  T2_COUNT                // 26
};
// F2: x flux (ID,IP,IU,IV,IW,IC), y flux (same six, y-normal frame), emfZ
enum { F2_X = 0, F2_Y = 6, F2_EMF = 12, F2_COUNT = 13 };

// Accessors (as in kernels_mhd3d.h): the numerics below are written once against small interfaces, so that the flat kernels
// (global SoA arrays, cell index = flat array index) and the LDS-tiled fused step of the HIP backend (hip/tiled_mhd2d.h) execute
// the same expressions in the same order.
//   trace inputs   q(v, m)  primitive variable v of cell m;  ua(m), ub(m)  low-face Bx / By of cell m;  sj()  row stride
//   T accessor     get(slot, m), sj();   T writer  put(slot, value)  (bound to one cell)
//   F writer       put(comp, value) (bound to one cell);   F reader  get(comp, m), sj()
struct Trace2dInGlobal {
  const double* U; const double* Q; size_t N; unsigned sj_;
  RG_DEVFN double q(int v, unsigned m) const { return Q[(size_t)v * N + m]; }
  RG_DEVFN double ua(unsigned m) const { return U[(size_t)IA * N + m]; }
  RG_DEVFN double ub(unsigned m) const { return U[(size_t)IB * N + m]; }
  RG_DEVFN unsigned sj() const { return sj_; }
};
struct T2GlobalWrite {
  double* t; size_t N;
  RG_DEVFN void put(int slot, double v) const { t[(size_t)slot * N] = v; }
};
struct T2GlobalRead {
  const double* T; size_t N; unsigned sj_;
  RG_DEVFN double get(int slot, unsigned m) const { return T[m + (size_t)slot * N]; }
  RG_DEVFN unsigned sj() const { return sj_; }
};
struct F2GlobalWrite {
  double* f; size_t N;
  RG_DEVFN void put(int comp, double v) const { f[(size_t)comp * N] = v; }
};
struct F2GlobalRead {
  const double* F; size_t N; unsigned sj_;
  RG_DEVFN double get(int comp, unsigned m) const { return F[m + (size_t)comp * N]; }
  RG_DEVFN unsigned sj() const { return sj_; }
};

// slopes + CTU trace of one cell (known to be inside the trace range): m = its index in the input accessor's space, xPos its
// cell-centre x (rotating frame only)
template <class TIN, class TW>
RG_DEVFN void mhd_trace2d_at(const DevParams& g, const TIN& in, const TW& tw, double dtdx, double dtdy, unsigned m, double xPos) {
  const unsigned sj = in.sj();
  const double st = g.slope_type;

  // Ez at the four corners of the cell: corner (di,dj) averages the 4 cells around vertex (i+di, j+dj)
  double Ez[2][2];
#pragma unroll
  for (int di = 0; di < 2; ++di)
#pragma unroll
    for (int dj = 0; dj < 2; ++dj) {
      const unsigned o = m + di + dj * sj;
      const double u = 0.25 * (in.q(IU, o - 1 - sj) + in.q(IU, o - 1) + in.q(IU, o - sj) + in.q(IU, o));
      const double v = 0.25 * (in.q(IV, o - 1 - sj) + in.q(IV, o - 1) + in.q(IV, o - sj) + in.q(IV, o));
      const double A = 0.5 * (in.ua(o - sj) + in.ua(o));
      const double B = 0.5 * (in.ub(o - 1) + in.ub(o));
      Ez[di][dj] = u * B - v * A;
    }
  const double ELL = Ez[0][0], ELR = Ez[0][1], ERL = Ez[1][0];

  double q[8], dx_[8], dy_[8];
#pragma unroll
  for (int v = 0; v < 8; ++v) {
    q[v] = in.q(v, m);
    if (st == 0) { dx_[v] = 0.0; dy_[v] = 0.0; }
    else if (st == 3) {
      double lo = q[v], hi = q[v];
#pragma unroll
      for (int dj = -1; dj <= 1; ++dj)
#pragma unroll
        for (int di = -1; di <= 1; ++di) {
          const double nb = in.q(v, (unsigned)((int)m + di + dj * (int)sj));
          lo = (nb < lo) ? nb : lo;
          hi = (nb > hi) ? nb : hi;
        }
      const double dfx = 0.5 * (in.q(v, m + 1) - in.q(v, m - 1)), dfy = 0.5 * (in.q(v, m + sj) - in.q(v, m - sj));
      const double dlim = positivity_limiter(lo, hi, q[v], fabs(dfx) + fabs(dfy));
      dx_[v] = dlim * dfx * 0.5;
      dy_[v] = dlim * dfy * 0.5;
    } else {   // (dx_, dy_ hold HALF slopes: dev_numerics.h, tvd_half_slope)
      dx_[v] = tvd_half_slope(st, in.q(v, m - 1), q[v], in.q(v, m + 1));
      dy_[v] = tvd_half_slope(st, in.q(v, m - sj), q[v], in.q(v, m + sj));
    }
  }
  double r = q[ID], p = q[IP], u = q[IU], v = q[IV], w = q[IW], A = q[IA], B = q[IB], C = q[IC];
  double AL = in.ua(m), BL = in.ub(m);
  const double AR = in.ua(m + 1), BR = in.ub(m + sj);
  const double drx = dx_[ID], dpx = dx_[IP], dux = dx_[IU], dvx = dx_[IV], dwx = dx_[IW], dCx = dx_[IC], dBx = dx_[IB];
  const double dry = dy_[ID], dpy = dy_[IP], duy = dy_[IU], dvy = dy_[IV], dwy = dy_[IW], dCy = dy_[IC], dAy = dy_[IA];
  // transverse slopes of the low-face field (slope_unsplit_mhd_2d: slope type NOT capped)
  const double dALy = tvd_half_slope(st, in.ua(m - sj), AL, in.ua(m + sj));
  const double dBLx = tvd_half_slope(st, in.ub(m - 1), BL, in.ub(m + 1));
  const double dAx = 0.5 * (AR - AL), dBy = 0.5 * (BR - BL);
  const double gamma = g.gamma0;

  const double sr0 = (-u * drx - dux * r) * dtdx + (-v * dry - dvy * r) * dtdy;
  const rg_recip_t inv_r = rg_recip(r);
  const double su0 = (-u * dux - rg_div(dpx, inv_r) - rg_div(B * dBx, inv_r) - rg_div(C * dCx, inv_r)) * dtdx + (-v * duy + rg_div(B * dAy, inv_r)) * dtdy;
  const double sv0 = (-u * dvx + rg_div(A * dBx, inv_r)) * dtdx + (-v * dvy - rg_div(dpy, inv_r) - rg_div(A * dAy, inv_r) - rg_div(C * dCy, inv_r)) * dtdy;
  const double sw0 = (-u * dwx + rg_div(A * dCx, inv_r)) * dtdx + (-v * dwy + rg_div(B * dCy, inv_r)) * dtdy;
  const double sp0 = (-u * dpx - dux * gamma * p) * dtdx + (-v * dpy - dvy * gamma * p) * dtdy;
  const double sA0 = (u * dBy + B * duy - v * dAy - A * dvy) * dtdy;
  const double sB0 = (-u * dBx - B * dux + v * dAx + A * dvx) * dtdx;
  double sC0 = (w * dAx + A * dwx - u * dCx - C * dux) * dtdx + (-v * dCy - C * dvy + w * dBy + B * dwy) * dtdy;
  if (g.Omega0 > 0) {  // rotating frame (trace_mhd.h:213-217)
    const double shear = -1.5 * g.Omega0 * xPos;
    sC0 += (shear * dAx - 1.5 * g.Omega0 * A) * dtdx;
    sC0 += shear * dBy * dtdy;
  }
  const double sAL0 = +(ELR - ELL) * 0.5 * dtdy;
  const double sBL0 = -(ERL - ELL) * 0.5 * dtdx;

  r = r + sr0; u = u + su0; v = v + sv0; w = w + sw0; p = p + sp0; A = A + sA0; B = B + sB0; C = C + sC0;
  AL = AL + sAL0; BL = BL + sBL0;

  tw.put(T2_R, r); tw.put(T2_P, p); tw.put(T2_U, u); tw.put(T2_V, v); tw.put(T2_W, w); tw.put(T2_A, A); tw.put(T2_B, B); tw.put(T2_C, C);
  tw.put(T2_AL, AL); tw.put(T2_BL, BL);
  tw.put(T2_DX + 0, drx); tw.put(T2_DX + 1, dpx); tw.put(T2_DX + 2, dux); tw.put(T2_DX + 3, dvx); tw.put(T2_DX + 4, dwx);
  tw.put(T2_DX + 5, dBx); tw.put(T2_DX + 6, dCx);
  tw.put(T2_DY + 0, dry); tw.put(T2_DY + 1, dpy); tw.put(T2_DY + 2, duy); tw.put(T2_DY + 3, dvy); tw.put(T2_DY + 4, dwy);
  tw.put(T2_DY + 5, dAy); tw.put(T2_DY + 6, dCy);
  tw.put(T2_DALY, dALy); tw.put(T2_DBLX, dBLx);
}

RG_DEVFN void mhd_trace2d_cell(const DevParams& g, const double* __restrict__ U, const double* __restrict__ Q,
                               double* __restrict__ T, double dtdx, double dtdy, unsigned idx) {
  const IJK c = unflatten(g, idx);
  const int lo = g.gw - 1;
  if (c.i < lo || c.i > g.isize - g.gw || c.j < lo || c.j > g.jsize - g.gw) return;
  const Trace2dInGlobal in = {U, Q, (size_t)g.ncell, g.sj};
  const T2GlobalWrite tw = {T + idx, (size_t)g.ncell};
  const double xPos = g.xMin + g.dx / 2 + (c.i - g.gw) * g.dx;
  mhd_trace2d_at(g, in, tw, dtdx, dtdy, idx, xPos);
}

// 2D trace floors: rho >= smallr, p >= smallp * rho (trace_mhd.h:251-252)
RG_DEVFN void floor2d(const DevParams& g, Prim8& s) {
  s.r = fmax(g.smallr, s.r);
  s.p = fmax(g.smallp * s.r, s.p);
}

// qm[D] (SIDE=+1) / qp[D] (SIDE=-1) of cell m, in the face-normal frame (trace_mhd.h:242-288).  gm = GLOBAL flat index of the
// cell (only the per-cell gravity field reads it)
template <int D, int SIDE, bool GF, class TA>
RG_DEVFN Prim8 face_state2d(const DevParams& g, const TA& T, unsigned m, unsigned gm) {
  const unsigned sD = (D == XD) ? 1u : T.sj();
  const int S = (D == XD) ? T2_DX : T2_DY;
  const double s = (double)SIDE;
  Prim8 o;
  o.r = T.get(T2_R, m) + s * T.get(S + 0, m);
  o.p = T.get(T2_P, m) + s * T.get(S + 1, m);
  double u = T.get(T2_U, m) + s * T.get(S + 2, m);
  double v = T.get(T2_V, m) + s * T.get(S + 3, m);
  o.w = T.get(T2_W, m) + s * T.get(S + 4, m);
  const int TF = (D == XD) ? T2_AL : T2_BL;
  const double bn = (SIDE > 0) ? T.get(TF, m + sD) : T.get(TF, m);
  const double bt = ((D == XD) ? T.get(T2_B, m) : T.get(T2_A, m)) + s * T.get(S + 5, m);
  o.c = T.get(T2_C, m) + s * T.get(S + 6, m);
  if (D == XD) { o.u = u; o.v = v; } else { o.u = v; o.v = u; }
  // implementation version 0: the reference adds the gravity predictor AFTER the swap into the face-normal frame
  // (mhd_godunov_unsplit_cpu_v0.cpp:177-179, 388-390, then 500-512), so on y faces g_x lands on v and g_y on u
  if (GF || g.grav_on) {
    double gx, gy, gz;
    half_dt_gravity<GF>(g, gm, gx, gy, gz);
    o.u += gx; o.v += gy;
  }
  o.a = bn; o.b = bt;
  floor2d(g, o);
  return o;
}

// qEdge of cell m at corner (SX,SY) (trace_mhd.h:291-337), grid frame (= the edge frame of emfZ)
template <int SX, int SY, bool GF, class TA>
RG_DEVFN Prim8 edge_state2d(const DevParams& g, const TA& T, unsigned m, unsigned gm) {
  const double sx = (double)SX, sy = (double)SY;
  Prim8 o;
  o.r = T.get(T2_R, m) + (sx * T.get(T2_DX + 0, m) + sy * T.get(T2_DY + 0, m));
  o.p = T.get(T2_P, m) + (sx * T.get(T2_DX + 1, m) + sy * T.get(T2_DY + 1, m));
  o.u = T.get(T2_U, m) + (sx * T.get(T2_DX + 2, m) + sy * T.get(T2_DY + 2, m));
  o.v = T.get(T2_V, m) + (sx * T.get(T2_DX + 3, m) + sy * T.get(T2_DY + 3, m));
  if (GF || g.grav_on) {   // (mhd_godunov_unsplit_cpu_v0.cpp:514-524)
    double gx, gy, gz;
    half_dt_gravity<GF>(g, gm, gx, gy, gz);
    o.u += gx; o.v += gy;
  }
  o.w = T.get(T2_W, m) + (sx * T.get(T2_DX + 4, m) + sy * T.get(T2_DY + 4, m));
  o.c = T.get(T2_C, m) + (sx * T.get(T2_DX + 6, m) + sy * T.get(T2_DY + 6, m));
  const unsigned mx = (SX > 0) ? m + 1 : m;
  const unsigned my = (SY > 0) ? m + T.sj() : m;
  o.a = T.get(T2_AL, mx) + sy * T.get(T2_DALY, mx);
  o.b = T.get(T2_BL, my) + sx * T.get(T2_DBLX, my);
  floor2d(g, o);
  return o;
}

// The three Riemann problems at the low faces / the low corner of cell m (known to be in range); gm = its global flat index,
// xPos its cell-centre x.  WHAT: bit 0 = x flux, bit 1 = y flux, bit 2 = emfZ (the tiled kernel deals them to different waves)
enum { DO2_FX = 1, DO2_FY = 2, DO2_EMF = 4, DO2_ALL = 7 };
template <int WHAT, bool GF, class TA, class FW>
RG_DEVFN void mhd_flux2d_at(const DevParams& g, const TA& T, const FW& fw, unsigned m, unsigned gm, double xPos) {
  const unsigned sj = T.sj();
  double fl[8];
  // rotating frame (godunov_unsplit_rotating_cpu, 2D branch, MHDRunGodunov.cpp:2089-2434): the Bz fluxes get the
  // shear advection of the mean normal field left in the states by the Riemann solver, emfZ its upwind term
  if (WHAT & DO2_FX) {
    Prim8 L = face_state2d<XD, +1, GF>(g, T, m - 1, gm - 1), R = face_state2d<XD, -1, GF>(g, T, m, gm);
#pragma unroll
    for (int v = 0; v < 8; ++v) fl[v] = 0.0;
    mhd_riemann(g, L, R, fl);
    if (g.rot) {
      const double shear_x = -1.5 * g.Omega0 * (xPos + xPos - g.dx);
      fl[IC] += shear_x * (L.a + R.a) / 2;
    }
    fw.put(F2_X + 0, fl[ID]); fw.put(F2_X + 1, fl[IP]); fw.put(F2_X + 2, fl[IU]);
    fw.put(F2_X + 3, fl[IV]); fw.put(F2_X + 4, fl[IW]); fw.put(F2_X + 5, fl[IC]);
  }
  if (WHAT & DO2_FY) {
    Prim8 L = face_state2d<YD, +1, GF>(g, T, m - sj, gm - g.sj), R = face_state2d<YD, -1, GF>(g, T, m, gm);
#pragma unroll
    for (int v = 0; v < 8; ++v) fl[v] = 0.0;
    mhd_riemann(g, L, R, fl);
    if (g.rot) {
      const double shear_y = -1.5 * g.Omega0 * xPos;
      fl[IC] += shear_y * (L.a + R.a) / 2;
    }
    fw.put(F2_Y + 0, fl[ID]); fw.put(F2_Y + 1, fl[IP]); fw.put(F2_Y + 2, fl[IU]);
    fw.put(F2_Y + 3, fl[IV]); fw.put(F2_Y + 4, fl[IW]); fw.put(F2_Y + 5, fl[IC]);
  }
  if (WHAT & DO2_EMF) {
    const Prim8 rt = edge_state2d<+1, +1, GF>(g, T, m - 1 - sj, gm - 1 - g.sj), rb = edge_state2d<+1, -1, GF>(g, T, m - 1, gm - 1);
    const Prim8 lt = edge_state2d<-1, +1, GF>(g, T, m - sj, gm - g.sj), lb = edge_state2d<-1, -1, GF>(g, T, m, gm);
    fw.put(F2_EMF, edge_emf<2>(g, rt, rb, lt, lb, xPos));
  }
}

template <bool GF>
RG_DEVFN void mhd_flux2d_cell(const DevParams& g, const double* __restrict__ T, double* __restrict__ F, unsigned idx) {
  const IJK c = unflatten(g, idx);
  if (c.i < g.gw || c.i > g.isize - g.gw || c.j < g.gw || c.j > g.jsize - g.gw) return;
  const double xPos = g.xMin + g.dx / 2 + (c.i - g.gw) * g.dx;
  const T2GlobalRead ta = {T, (size_t)g.ncell, g.sj};
  const F2GlobalWrite fw = {F + idx, (size_t)g.ncell};
  mhd_flux2d_at<DO2_ALL, GF>(g, ta, fw, idx, idx, xPos);
}

// Conservative + CT update of one cell from its old state u[8] (in: old, out: new).  fm = the cell's index in the flux
// accessor's space, gm its global flat index.  interior: the cell takes the six flux contributions; ct: it lies in the CT range
// [gw, size - gw] of both directions.  (The reference's 2D update has no guards -- it also scribbles on ghost cells that the
// next ghost fill overwrites; only interior cells and the CT range are reproduced.)
template <bool GF, class FA>
RG_DEVFN void mhd_update2d_at(const DevParams& g, const RotCoef rc, const FA& F, double* u, double rho_old, double dt, double dtdx,
                              double dtdy, unsigned fm, unsigned gm, bool interior, bool ct) {
  const unsigned sj = F.sj();
  if (interior) {
    double f[6];
#define RG_LOADF2(base, off) _Pragma("unroll") for (int v = 0; v < 6; ++v) f[v] = F.get((base) + v, fm + (off))
    if (!g.rot) {
      RG_LOADF2(F2_X, 0);
      u[ID] += f[0] * dtdx; u[IP] += f[1] * dtdx; u[IU] += f[2] * dtdx; u[IV] += f[3] * dtdx; u[IW] += f[4] * dtdx; u[IC] += f[5] * dtdx;
      RG_LOADF2(F2_Y, 0);  // y-normal frame: f[2] = y momentum flux, f[3] = x momentum flux
      u[ID] += f[0] * dtdy; u[IP] += f[1] * dtdy; u[IU] += f[3] * dtdy; u[IV] += f[2] * dtdy; u[IW] += f[4] * dtdy; u[IC] += f[5] * dtdy;
      RG_LOADF2(F2_X, 1);
      u[ID] -= f[0] * dtdx; u[IP] -= f[1] * dtdx; u[IU] -= f[2] * dtdx; u[IV] -= f[3] * dtdx; u[IW] -= f[4] * dtdx; u[IC] -= f[5] * dtdx;
      RG_LOADF2(F2_Y, sj);
      u[ID] -= f[0] * dtdy; u[IP] -= f[1] * dtdy; u[IU] -= f[3] * dtdy; u[IV] -= f[2] * dtdy; u[IW] -= f[4] * dtdy; u[IC] -= f[5] * dtdy;
    } else {
      // rotating frame (MHDRunGodunov.cpp:2253-2306): Coriolis on the old momenta before any flux reaches the cell,
      // then the fluxes with the (alpha1, alpha2) mixing of the two in-plane momentum components
      const rg_recip_t inv_l = rg_recip(1.0 + rc.lambda);
      const double dsx = rg_div(2.0 * g.Omega0 * dt * u[IV], inv_l);
      const double dsy = rg_div(-0.5 * g.Omega0 * dt * u[IU], inv_l);
      u[IU] = u[IU] * rc.ratio + dsx;
      u[IV] = u[IV] * rc.ratio + dsy;
      const double a1 = rc.alpha1, a2 = rc.alpha2;
      RG_LOADF2(F2_X, 0);
      u[ID] += f[0] * dtdx; u[IP] += f[1] * dtdx;
      u[IU] += (a1 * f[2] + a2 * f[3]) * dtdx; u[IV] += (a1 * f[3] - 0.25 * a2 * f[2]) * dtdx;
      u[IW] += f[4] * dtdx; u[IC] += f[5] * dtdx;
      RG_LOADF2(F2_Y, 0);
      u[ID] += f[0] * dtdy; u[IP] += f[1] * dtdy;
      u[IU] += (a1 * f[3] + a2 * f[2]) * dtdy; u[IV] += (a1 * f[2] - 0.25 * a2 * f[3]) * dtdy;
      u[IW] += f[4] * dtdy; u[IC] += f[5] * dtdy;
      RG_LOADF2(F2_X, 1);
      u[ID] -= f[0] * dtdx; u[IP] -= f[1] * dtdx;
      u[IU] -= (a1 * f[2] + a2 * f[3]) * dtdx; u[IV] -= (a1 * f[3] - 0.25 * a2 * f[2]) * dtdx;
      u[IW] -= f[4] * dtdx; u[IC] -= f[5] * dtdx;
      RG_LOADF2(F2_Y, sj);
      u[ID] -= f[0] * dtdy; u[IP] -= f[1] * dtdy;
      u[IU] -= (a1 * f[3] + a2 * f[2]) * dtdy; u[IV] -= (a1 * f[2] - 0.25 * a2 * f[3]) * dtdy;
      u[IW] -= f[4] * dtdy; u[IC] -= f[5] * dtdy;
    }
#undef RG_LOADF2
    if (GF || g.grav_on) {  // momentum source (mhd_godunov_unsplit_cpu_v0.cpp:616-618)
      const double rho_sum = rho_old + u[ID];
      double gx, gy, gz;
      half_dt_gravity<GF>(g, gm, gx, gy, gz);
      u[IU] += gx * rho_sum;
      u[IV] += gy * rho_sum;
    }
  }
  if (ct) {
    u[IA] += (F.get(F2_EMF, fm + sj) - F.get(F2_EMF, fm)) * dtdy;
    u[IB] -= (F.get(F2_EMF, fm + 1) - F.get(F2_EMF, fm)) * dtdx;
  }
}

// 2D value of mhd_invdt_cell for an interior cell whose NEW state is u[8]: the new field on its two high faces belongs to the +1
// neighbours, whose CT update is repeated here from the same emf values (ua_x1 / ub_y1: their OLD face fields)
template <class FA>
RG_DEVFN double mhd_invdt2d_new(const DevParams& g, const FA& F, const double* u, double ua_x1, double ub_y1, double dtdx, double dtdy,
                                unsigned fm) {
  const unsigned sj = F.sj();
  unsigned m = fm + 1;
  const double bnx = ua_x1 + (F.get(F2_EMF, m + sj) - F.get(F2_EMF, m)) * dtdy;
  m = fm + sj;
  const double bny = ub_y1 - (F.get(F2_EMF, m + 1) - F.get(F2_EMF, m)) * dtdx;
  const Prim8 q = mhd_prim(g, u, bnx, bny, 0.0, 0.0);
  double sx, sy, sz;
  info_speeds(g, q, sx, sy, sz);
  return sx / g.dx + sy / g.dy;
}

template <bool GF>
RG_DEVFN void mhd_update2d_cell(const DevParams& g, const RotCoef rc, const double* __restrict__ Uold,
                                double* __restrict__ Unew, const double* __restrict__ F, double dt, double dtdx, double dtdy,
                                unsigned idx, unsigned long long* dt_slots = 0) {
  const IJK c = unflatten(g, idx);
  const size_t N = g.ncell;
  const unsigned sj = g.sj;
  const int gw = g.gw;
  double u[8];
#pragma unroll
  for (int v = 0; v < 8; ++v) u[v] = Uold[idx + v * N];
  const bool in_i = c.i >= gw && c.i < g.isize - gw, in_j = c.j >= gw && c.j < g.jsize - gw;
  const bool ct = c.i >= gw && c.i <= g.isize - gw && c.j >= gw && c.j <= g.jsize - gw;
  const F2GlobalRead fa = {F, N, sj};
  mhd_update2d_at<GF>(g, rc, fa, u, u[ID], dt, dtdx, dtdy, idx, idx, in_i && in_j, ct);
  // dt_slots != 0: the CFL scan of the NEW state rides along (as in mhd_update3d_cell)
  if (dt_slots) {   // (all lanes of the wave: the maximum is formed wave-wide before it goes to a slot)
    double inv = 0.0;
    if (in_i && in_j) inv = mhd_invdt2d_new(g, fa, u, Uold[idx + 1 + IA * N], Uold[idx + sj + IB * N], dtdx, dtdy, idx);
    rgpu::rg_slot_max_wave(dt_slots + ((idx >> 6) & (rgpu::RG_DT_SLOTS - 1)), inv);
  }
#pragma unroll
  for (int v = 0; v < 8; ++v) Unew[idx + v * N] = u[v];
}

}  // namespace rgpu_dev
