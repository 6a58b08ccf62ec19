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

RG_DEVFN void mhd_trace2d_cell(const DevParams& g, const double* __restrict__ U, const double* __restrict__ Q,
                               double* __restrict__ T, double dtdx, double dtdy, unsigned idx) {
  const IJK c = unflatten(g, idx);
  const int lo = g.gw - 1;
  if (c.i < lo || c.i > g.isize - g.gw || c.j < lo || c.j > g.jsize - g.gw) return;
  const size_t N = g.ncell;
  const unsigned sj = g.sj;
  const double st = g.slope_type;
  const double* Qu = Q + IU * N; const double* Qv = Q + IV * N;
  const double* Ua = U + IA * N; const double* Ub = U + IB * N;

  // Ez at the four corners of the cell: corner (di,dj) averages the 4 cells around vertex (i+di, j+dj)
  double Ez[2][2];
#pragma unroll
  for (int di = 0; di < 2; ++di)
#pragma unroll
    for (int dj = 0; dj < 2; ++dj) {
      const unsigned o = idx + di + dj * sj;
      const double u = 0.25 * (Qu[o - 1 - sj] + Qu[o - 1] + Qu[o - sj] + Qu[o]);
      const double v = 0.25 * (Qv[o - 1 - sj] + Qv[o - 1] + Qv[o - sj] + Qv[o]);
      const double A = 0.5 * (Ua[o - sj] + Ua[o]);
      const double B = 0.5 * (Ub[o - 1] + Ub[o]);
      Ez[di][dj] = u * B - v * A;
    }
  const double ELL = Ez[0][0], ELR = Ez[0][1], ERL = Ez[1][0];

  double q[8], dx_[8], dy_[8];
#pragma unroll
  for (int v = 0; v < 8; ++v) {
    const double* Qc = Q + v * N;
    q[v] = Qc[idx];
    if (st == 0) { dx_[v] = 0.0; dy_[v] = 0.0; }
    else if (st == 3) {
      double lo = q[v], hi = q[v];
#pragma unroll
      for (int dj = -1; dj <= 1; ++dj)
#pragma unroll
        for (int di = -1; di <= 1; ++di) {
          const double nb = Qc[(unsigned)((int)idx + di + dj * (int)sj)];
          lo = (nb < lo) ? nb : lo;
          hi = (nb > hi) ? nb : hi;
        }
      const double dfx = 0.5 * (Qc[idx + 1] - Qc[idx - 1]), dfy = 0.5 * (Qc[idx + sj] - Qc[idx - sj]);
      const double dlim = positivity_limiter(lo, hi, q[v], fabs(dfx) + fabs(dfy));
      dx_[v] = dlim * dfx;
      dy_[v] = dlim * dfy;
    } else {
      dx_[v] = tvd_slope(st, Qc[idx - 1], q[v], Qc[idx + 1]);
      dy_[v] = tvd_slope(st, Qc[idx - sj], q[v], Qc[idx + sj]);
    }
  }
  double r = q[ID], p = q[IP], u = q[IU], v = q[IV], w = q[IW], A = q[IA], B = q[IB], C = q[IC];
  double AL = Ua[idx], BL = Ub[idx];
  const double AR = Ua[idx + 1], BR = Ub[idx + sj];
  const double drx = dx_[ID] * 0.5, dpx = dx_[IP] * 0.5, dux = dx_[IU] * 0.5, dvx = dx_[IV] * 0.5, dwx = dx_[IW] * 0.5,
               dCx = dx_[IC] * 0.5, dBx = dx_[IB] * 0.5;
  const double dry = dy_[ID] * 0.5, dpy = dy_[IP] * 0.5, duy = dy_[IU] * 0.5, dvy = dy_[IV] * 0.5, dwy = dy_[IW] * 0.5,
               dCy = dy_[IC] * 0.5, dAy = dy_[IA] * 0.5;
  // transverse slopes of the low-face field (slope_unsplit_mhd_2d: slope type NOT capped)
  const double dALy = 0.5 * tvd_slope(st, Ua[idx - sj], AL, Ua[idx + sj]);
  const double dBLx = 0.5 * tvd_slope(st, Ub[idx - 1], BL, Ub[idx + 1]);
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
    const double xPos = g.xMin + g.dx / 2 + (c.i - g.gw) * g.dx;
    const double shear = -1.5 * g.Omega0 * xPos;
    sC0 += (shear * dAx - 1.5 * g.Omega0 * A) * dtdx;
    sC0 += shear * dBy * dtdy;
  }
  const double sAL0 = +(ELR - ELL) * 0.5 * dtdy;
  const double sBL0 = -(ERL - ELL) * 0.5 * dtdx;

  r = r + sr0; u = u + su0; v = v + sv0; w = w + sw0; p = p + sp0; A = A + sA0; B = B + sB0; C = C + sC0;
  AL = AL + sAL0; BL = BL + sBL0;

  double* t = T + idx;
  t[T2_R * N] = r; t[T2_P * N] = p; t[T2_U * N] = u; t[T2_V * N] = v; t[T2_W * N] = w; t[T2_A * N] = A; t[T2_B * N] = B; t[T2_C * N] = C;
  t[T2_AL * N] = AL; t[T2_BL * N] = BL;
  t[(T2_DX + 0) * N] = drx; t[(T2_DX + 1) * N] = dpx; t[(T2_DX + 2) * N] = dux; t[(T2_DX + 3) * N] = dvx; t[(T2_DX + 4) * N] = dwx;
  t[(T2_DX + 5) * N] = dBx; t[(T2_DX + 6) * N] = dCx;
  t[(T2_DY + 0) * N] = dry; t[(T2_DY + 1) * N] = dpy; t[(T2_DY + 2) * N] = duy; t[(T2_DY + 3) * N] = dvy; t[(T2_DY + 4) * N] = dwy;
  t[(T2_DY + 5) * N] = dAy; t[(T2_DY + 6) * N] = dCy;
  t[T2_DALY * N] = dALy; t[T2_DBLX * N] = dBLx;
}

// 2D trace floors: rho >= smallr, p >= smallp * rho (trace_mhd.h:251-252)
RG_DEVFN void floor2d(const DevParams& g, Prim8& s) {
  s.r = fmax(g.smallr, s.r);
  s.p = fmax(g.smallp * s.r, s.p);
}

// qm[D] (SIDE=+1) / qp[D] (SIDE=-1) of cell m, in the face-normal frame (trace_mhd.h:242-288)
template <int D, int SIDE, bool GF>
RG_DEVFN Prim8 face_state2d(const DevParams& g, const double* __restrict__ T, unsigned m) {
  const size_t N = g.ncell;
  const unsigned sD = (D == XD) ? 1u : g.sj;
  const int S = (D == XD) ? T2_DX : T2_DY;
  const double* t = T + m;
  const double s = (double)SIDE;
  Prim8 o;
  o.r = t[T2_R * N] + s * t[(S + 0) * N];
  o.p = t[T2_P * N] + s * t[(S + 1) * N];
  double u = t[T2_U * N] + s * t[(S + 2) * N];
  double v = t[T2_V * N] + s * t[(S + 3) * N];
  o.w = t[T2_W * N] + s * t[(S + 4) * N];
  const int TF = (D == XD) ? T2_AL : T2_BL;
  const double bn = (SIDE > 0) ? T[(m + sD) + (size_t)TF * N] : t[TF * N];
  const double bt = ((D == XD) ? t[T2_B * N] : t[T2_A * N]) + s * t[(S + 5) * N];
  o.c = t[T2_C * N] + s * t[(S + 6) * N];
  if (D == XD) { o.u = u; o.v = v; } else { o.u = v; o.v = u; }
  // implementation version 0: the reference adds the gravity predictor AFTER the swap into the face-normal frame
  // (mhd_godunov_unsplit_cpu_v0.cpp:177-179, 388-390, then 500-512), so on y faces g_x lands on v and g_y on u
  if (GF || g.grav_on) {
    double gx, gy, gz;
    half_dt_gravity<GF>(g, m, gx, gy, gz);
    o.u += gx; o.v += gy;
  }
  o.a = bn; o.b = bt;
  floor2d(g, o);
  return o;
}

// qEdge of cell m at corner (SX,SY) (trace_mhd.h:291-337), grid frame (= the edge frame of emfZ)
template <int SX, int SY, bool GF>
RG_DEVFN Prim8 edge_state2d(const DevParams& g, const double* __restrict__ T, unsigned m) {
  const size_t N = g.ncell;
  const double* t = T + m;
  const double sx = (double)SX, sy = (double)SY;
  Prim8 o;
  o.r = t[T2_R * N] + (sx * t[(T2_DX + 0) * N] + sy * t[(T2_DY + 0) * N]);
  o.p = t[T2_P * N] + (sx * t[(T2_DX + 1) * N] + sy * t[(T2_DY + 1) * N]);
  o.u = t[T2_U * N] + (sx * t[(T2_DX + 2) * N] + sy * t[(T2_DY + 2) * N]);
  o.v = t[T2_V * N] + (sx * t[(T2_DX + 3) * N] + sy * t[(T2_DY + 3) * N]);
  if (GF || g.grav_on) {   // (mhd_godunov_unsplit_cpu_v0.cpp:514-524)
    double gx, gy, gz;
    half_dt_gravity<GF>(g, m, gx, gy, gz);
    o.u += gx; o.v += gy;
  }
  o.w = t[T2_W * N] + (sx * t[(T2_DX + 4) * N] + sy * t[(T2_DY + 4) * N]);
  o.c = t[T2_C * N] + (sx * t[(T2_DX + 6) * N] + sy * t[(T2_DY + 6) * N]);
  const unsigned mx = (SX > 0) ? m + 1 : m;
  const unsigned my = (SY > 0) ? m + g.sj : m;
  o.a = T[mx + (size_t)T2_AL * N] + sy * T[mx + (size_t)T2_DALY * N];
  o.b = T[my + (size_t)T2_BL * N] + sx * T[my + (size_t)T2_DBLX * N];
  floor2d(g, o);
  return o;
}

template <bool GF>
RG_DEVFN void mhd_flux2d_cell(const DevParams& g, const double* __restrict__ T, double* __restrict__ F, unsigned idx) {
  const IJK c = unflatten(g, idx);
  if (c.i < g.gw || c.i > g.isize - g.gw || c.j < g.gw || c.j > g.jsize - g.gw) return;
  const size_t N = g.ncell;
  const unsigned sj = g.sj;
  double fl[8];
  // rotating frame (godunov_unsplit_rotating_cpu, 2D branch, MHDRunGodunov.cpp:2089-2434): the Bz fluxes get the
  // shear advection of the mean normal field left in the states by the Riemann solver, emfZ its upwind term
  const double xPos = g.xMin + g.dx / 2 + (c.i - g.gw) * g.dx;
  {
    Prim8 L = face_state2d<XD, +1, GF>(g, T, idx - 1), R = face_state2d<XD, -1, GF>(g, T, idx);
#pragma unroll
    for (int v = 0; v < 8; ++v) fl[v] = 0.0;
    mhd_riemann(g, L, R, fl);
    if (g.rot) {
      const double shear_x = -1.5 * g.Omega0 * (xPos + xPos - g.dx);
      fl[IC] += shear_x * (L.a + R.a) / 2;
    }
    F[idx + (size_t)(F2_X + 0) * N] = fl[ID]; F[idx + (size_t)(F2_X + 1) * N] = fl[IP]; F[idx + (size_t)(F2_X + 2) * N] = fl[IU];
    F[idx + (size_t)(F2_X + 3) * N] = fl[IV]; F[idx + (size_t)(F2_X + 4) * N] = fl[IW]; F[idx + (size_t)(F2_X + 5) * N] = fl[IC];
  }
  {
    Prim8 L = face_state2d<YD, +1, GF>(g, T, idx - sj), R = face_state2d<YD, -1, GF>(g, T, idx);
#pragma unroll
    for (int v = 0; v < 8; ++v) fl[v] = 0.0;
    mhd_riemann(g, L, R, fl);
    if (g.rot) {
      const double shear_y = -1.5 * g.Omega0 * xPos;
      fl[IC] += shear_y * (L.a + R.a) / 2;
    }
    F[idx + (size_t)(F2_Y + 0) * N] = fl[ID]; F[idx + (size_t)(F2_Y + 1) * N] = fl[IP]; F[idx + (size_t)(F2_Y + 2) * N] = fl[IU];
    F[idx + (size_t)(F2_Y + 3) * N] = fl[IV]; F[idx + (size_t)(F2_Y + 4) * N] = fl[IW]; F[idx + (size_t)(F2_Y + 5) * N] = fl[IC];
  }
  {
    const Prim8 rt = edge_state2d<+1, +1, GF>(g, T, idx - 1 - sj), rb = edge_state2d<+1, -1, GF>(g, T, idx - 1);
    const Prim8 lt = edge_state2d<-1, +1, GF>(g, T, idx - sj), lb = edge_state2d<-1, -1, GF>(g, T, idx);
    F[idx + (size_t)F2_EMF * N] = edge_emf<2>(g, rt, rb, lt, lb, xPos);
  }
}

// The reference's 2D update has no guards (it also scribbles on ghost cells that the next ghost fill
// overwrites); only interior cells and the CT range are reproduced, everything else is copied.
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
  if (in_i && in_j) {
    double f[6];
#define RG_LOADF2(base, off) _Pragma("unroll") for (int v = 0; v < 6; ++v) f[v] = F[(idx + (off)) + (size_t)((base) + v) * N]
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
      const double rho_sum = Uold[idx + ID * N] + u[ID];
      double gx, gy, gz;
      half_dt_gravity<GF>(g, idx, gx, gy, gz);
      u[IU] += gx * rho_sum;
      u[IV] += gy * rho_sum;
    }
  }
  if (c.i >= gw && c.i <= g.isize - gw && c.j >= gw && c.j <= g.jsize - gw) {
    const double* e = F + (size_t)F2_EMF * N;
    u[IA] += (e[idx + sj] - e[idx]) * dtdy;
    u[IB] -= (e[idx + 1] - e[idx]) * dtdx;
  }
  // dt_slots != 0: the CFL scan of the NEW state rides along (as in mhd_update3d_cell): the new field on the two high faces
  // belongs to the +1 neighbours, whose CT update is repeated here from the same emf values; 2D value of mhd_invdt_cell
  if (dt_slots) {   // (all lanes of the wave: the maximum is formed wave-wide before it goes to a slot)
    double inv = 0.0;
    if (in_i && in_j) {
      const double* e = F + (size_t)F2_EMF * N;
      unsigned m = idx + 1;
      const double bnx = Uold[m + IA * N] + (e[m + sj] - e[m]) * dtdy;
      m = idx + sj;
      const double bny = Uold[m + IB * N] - (e[m + 1] - e[m]) * dtdx;
      const Prim8 q = mhd_prim(g, u, bnx, bny, 0.0, 0.0);
      double sx, sy, sz;
      info_speeds(g, q, sx, sy, sz);
      inv = sx / g.dx + sy / g.dy;
    }
    rgpu::rg_slot_max_wave(dt_slots + ((idx >> 6) & (rgpu::RG_DT_SLOTS - 1)), inv);
  }
#pragma unroll
  for (int v = 0; v < 8; ++v) Unew[idx + v * N] = u[v];
}

}  // namespace rgpu_dev
