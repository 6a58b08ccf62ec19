// kernels_hydro.h -- per-cell bodies of the hydro unsplit step ("unsplitVersion 1" and, through the order of the
// update, "unsplitVersion 2"), 2D (NV=4) and 3D (NV=5).
//   hydro_prim_cell    U         -> Q  (NV)          convertToPrimitives   HydroRunGodunov.cpp:4133-4262
//   hydro_trace_cell   Q         -> TH (NV*(1+ND))   slopes + trace        HydroRunGodunov.cpp:2454-2509, 2666-2748
//   hydro_flux_cell    TH        -> FH (NV*ND)       Riemann at low faces  HydroRunGodunov.cpp:2525-2565, 2757-2822
//   hydro_update_cell  Uold,FH   -> Unew             gather form of the scatter update :2574-2607, :2831-2895
// TH is the compact traced state: the time-advanced cell state and the limited half slopes; the face states
// qm/qp = state +/- half slope (+ floors) are rebuilt in the flux kernel (trace.h:384-412, 610-659).
#pragma once
#include "kernels_mhd3d.h"

namespace rgpu_dev {

template <int NV>
RG_DEVFN void hydro_prim_cell(const DevParams& g, const double* __restrict__ U, double* __restrict__ Q, unsigned idx) {
  const size_t N = g.ncell;
  double u[NV], q[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) u[v] = U[idx + v * N];
  hydro_prim<NV>(g, u, q);
#pragma unroll
  for (int v = 0; v < NV; ++v) Q[idx + v * N] = q[v];
}

template <int ND, int NV>
RG_DEVFN void hydro_trace_cell(const DevParams& g, const double* __restrict__ Q, double* __restrict__ T, double dtdx,
                               double dtdy, double dtdz, unsigned idx) {
  const IJK c = unflatten(g, idx);
  if (c.i < 1 || c.i >= g.isize - 1 || c.j < 1 || c.j >= g.jsize - 1) return;
  if (ND == 3 && (c.k < 1 || c.k >= g.ksize - 1)) return;
  const size_t N = g.ncell;
  const unsigned strd[3] = {1u, g.sj, g.sk};
  const double st = g.slope_type;
  double q[NV], h[ND][NV];  // h = HALF slopes
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const double* Qc = Q + v * N;
    q[v] = Qc[idx];
#pragma unroll
    for (int d = 0; d < ND; ++d) {
      const double qm = Qc[idx - strd[d]], qp = Qc[idx + strd[d]];
      double s;   // the reference's slope x 0.5 (trace.h:350-360), formed as a half slope (dev_numerics.h: tvd_half_slope)
      if (st == 0) s = 0.0;
      else if (ND == 3 && st == 1) s = minmod_half_slope(qm, q[v], qp);   // slope_unsplit_3d type 1 (slope.h:351-384)
      else s = tvd_half_slope(st, qm, q[v], qp);
      h[d][v] = s;
    }
  }
  double r = q[ID], p = q[IP], u = q[IU], v = q[IV], w = (NV == 5) ? q[IW] : 0.0;
  const double gamma = g.gamma0;
  const double drx = h[0][ID], dpx = h[0][IP], dux = h[0][IU], dvx = h[0][IV];
  const double dry = h[1][ID], dpy = h[1][IP], duy = h[1][IU], dvy = h[1][IV];
  double sr0, su0, sv0, sw0 = 0.0, sp0;
  const rg_recip_t inv_r = rg_recip(r);
  if (ND == 2) {
    sr0 = (-u * drx - dux * r) * dtdx + (-v * dry - dvy * r) * dtdy;
    su0 = (-u * dux - rg_div(dpx, inv_r)) * dtdx + (-v * duy) * dtdy;
    sv0 = (-u * dvx) * dtdx + (-v * dvy - rg_div(dpy, inv_r)) * dtdy;
    sp0 = (-u * dpx - dux * gamma * p) * dtdx + (-v * dpy - dvy * gamma * p) * dtdy;
  } else {
    const double dwx = h[0][NV - 1], dwy = h[1][NV - 1];
    const double drz = h[ND - 1][ID], dpz = h[ND - 1][IP], duz = h[ND - 1][IU], dvz = h[ND - 1][IV], dwz = h[ND - 1][NV - 1];
    sr0 = (-u * drx - dux * r) * dtdx + (-v * dry - dvy * r) * dtdy + (-w * drz - dwz * r) * dtdz;
    su0 = (-u * dux - rg_div(dpx, inv_r)) * dtdx + (-v * duy) * dtdy + (-w * duz) * dtdz;
    sv0 = (-u * dvx) * dtdx + (-v * dvy - rg_div(dpy, inv_r)) * dtdy + (-w * dvz) * dtdz;
    sw0 = (-u * dwx) * dtdx + (-v * dwy) * dtdy + (-w * dwz - rg_div(dpz, inv_r)) * dtdz;
    sp0 = (-u * dpx - dux * gamma * p) * dtdx + (-v * dpy - dvy * gamma * p) * dtdy + (-w * dpz - dwz * gamma * p) * dtdz;
  }
  r = r + sr0; u = u + su0; v = v + sv0; w = w + sw0; p = p + sp0;
  double* t = T + idx;
  t[ID * N] = r; t[IP * N] = p; t[IU * N] = u; t[IV * N] = v;
  if (NV == 5) t[IW * N] = w;
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int n = 0; n < NV; ++n) t[(size_t)(NV * (1 + d) + n) * N] = h[d][n];
}

// qm[D] (SIDE=+1) / qp[D] (SIDE=-1) of cell m in the face-normal frame, with the floors of trace.h:388-389
template <int D, int SIDE, int NV, bool GF>
RG_DEVFN void hydro_face_state(const DevParams& g, const double* __restrict__ T, unsigned m, double* o) {
  const size_t N = g.ncell;
  const double* t = T + m;
  const double s = (double)SIDE;
  double qv[NV];
#pragma unroll
  for (int n = 0; n < NV; ++n) qv[n] = t[(size_t)n * N] + s * t[(size_t)(NV * (1 + D) + n) * N];
  qv[ID] = fmax(g.smallr, qv[ID]);
  qv[IP] = fmax(g.smallp * qv[ID], qv[IP]);
  if (GF || g.grav_on) {  // gravity predictor on the traced state (HydroRunGodunov.cpp:2485-2497, 2705-2734)
    double gx, gy, gz;
    half_dt_gravity<GF>(g, m, gx, gy, gz);
    qv[IU] += gx;
    qv[IV] += gy;
    if (NV == 5) qv[NV - 1] += gz;
  }
  const int swp = (D == 0) ? IU : (D == 1) ? IV : IW;  // swap IU with the normal velocity
#pragma unroll
  for (int n = 0; n < NV; ++n) o[n] = qv[(n == IU) ? swp : (n == swp) ? IU : n];
}

template <int ND, int NV, bool GF>
RG_DEVFN void hydro_flux_cell(const DevParams& g, const double* __restrict__ T, double* __restrict__ F, unsigned idx) {
  const IJK c = unflatten(g, idx);
  if (c.i < g.gw || c.i > g.isize - g.gw || c.j < g.gw || c.j > g.jsize - g.gw) return;
  if (ND == 3 && (c.k < g.gw || c.k > g.ksize - g.gw)) return;
  const size_t N = g.ncell;
  double ql[NV], qr[NV], fl[NV];
  {
    hydro_face_state<0, +1, NV, GF>(g, T, idx - 1, ql);
    hydro_face_state<0, -1, NV, GF>(g, T, idx, qr);
#pragma unroll
    for (int n = 0; n < NV; ++n) fl[n] = 0.0;
    hydro_riemann<NV>(g, ql, qr, fl);
#pragma unroll
    for (int n = 0; n < NV; ++n) F[idx + (size_t)n * N] = fl[n];
  }
  {
    hydro_face_state<1, +1, NV, GF>(g, T, idx - g.sj, ql);
    hydro_face_state<1, -1, NV, GF>(g, T, idx, qr);
#pragma unroll
    for (int n = 0; n < NV; ++n) fl[n] = 0.0;
    hydro_riemann<NV>(g, ql, qr, fl);
#pragma unroll
    for (int n = 0; n < NV; ++n) F[idx + (size_t)(NV + n) * N] = fl[n];
  }
  if (ND == 3) {
    hydro_face_state<2, +1, NV, GF>(g, T, idx - g.sk, ql);
    hydro_face_state<2, -1, NV, GF>(g, T, idx, qr);
#pragma unroll
    for (int n = 0; n < NV; ++n) fl[n] = 0.0;
    hydro_riemann<NV>(g, ql, qr, fl);
#pragma unroll
    for (int n = 0; n < NV; ++n) F[idx + (size_t)(2 * NV + n) * N] = fl[n];
  }
}

template <int ND, int NV, bool GF>
RG_DEVFN void hydro_update_cell(const DevParams& g, const double* __restrict__ Uold, double* __restrict__ Unew,
                                const double* __restrict__ F, double dtdx, double dtdy, double dtdz, unsigned idx,
                                unsigned long long* dt_slots = 0) {
  const IJK c = unflatten(g, idx);
  const size_t N = g.ncell;
  const int gw = g.gw;
  double u[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) u[v] = Uold[idx + v * N];
  bool inner = c.i >= gw && c.i < g.isize - gw && c.j >= gw && c.j < g.jsize - gw;
  if (ND == 3) inner = inner && c.k >= gw && c.k < g.ksize - gw;
  if (inner) {
    const unsigned strd[3] = {1u, g.sj, g.sk};
    const double dtd[3] = {dtdx, dtdy, dtdz};
    // unsplitVersion 1: own low faces first (+), in x,y,z order; then the high faces (-), in x,y,z order.
    // unsplitVersion 2 (direction-wise sweeps of the reference, HydroRunGodunov.cpp:2955-3849): +x, -x, +y, -y, +z, -z.
    auto apply = [&](int d, int pass) {
      const unsigned o = idx + (pass ? strd[d] : 0u);
      const int swp = (d == 0) ? IU : (d == 1) ? IV : IW;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int vs = (v == IU) ? swp : (v == swp) ? IU : v;   // back from the face-normal frame
        const double f = F[o + (size_t)(d * NV + vs) * N] * dtd[d];
        if (pass == 0) u[v] += f; else u[v] -= f;
      }
    };
    if (!g.dirwise_update) {
#pragma unroll
      for (int pass = 0; pass < 2; ++pass)
#pragma unroll
        for (int d = 0; d < ND; ++d) apply(d, pass);
    } else {
#pragma unroll
      for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) apply(d, pass);
    }
    if (GF || g.grav_on) {  // momentum source (compute_gravity_source_term, HydroRunBase.cpp:1925-1985); energy untouched
      const double rho_sum = Uold[idx + ID * N] + u[ID];
      double gx, gy, gz;
      half_dt_gravity<GF>(g, idx, gx, gy, gz);
      u[IU] += gx * rho_sum;
      u[IV] += gy * rho_sum;
      if (NV == 5) u[NV - 1] += gz * rho_sum;
    }
  }
  if (dt_slots) {   // the CFL scan of the new state rides along (hydro_invdt_cell on the cell just updated); all lanes of the wave
    double inv = 0.0;
    if (inner) {
      double q[NV];
      const double cs = hydro_prim<NV>(g, u, q);
      inv = (cs + fabs(q[IU])) / g.dx + (cs + fabs(q[IV])) / g.dy;
      if (NV == 5) inv = (cs + fabs(q[IU])) / g.dx + (cs + fabs(q[IV])) / g.dy + (cs + fabs(q[IW])) / g.dz;
    }
    rgpu::rg_slot_max_wave(dt_slots + ((idx >> 6) & (rgpu::RG_DT_SLOTS - 1)), inv);
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) Unew[idx + v * N] = u[v];
}

}  // namespace rgpu_dev
