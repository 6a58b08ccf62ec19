// kernels_mhd3d.h -- per-cell bodies of the 3D MHD unsplit step (plain and rotating / shearing box).
//
// Pipeline (one launch each, flat 1D grid over the ghost-inclusive array, x fastest => fully coalesced SoA
// loads; neighbour re-reads are served by L2 / Infinity Cache):
//   mhd_prim_cell      U            -> Q   (8)    convertToPrimitives        MHDRunGodunov.cpp:519-560
//   mhd_elec_cell      U,Q          -> E   (3)    edge electric field        mhd_godunov_unsplit_cpu_v3.cpp:36-101
//                                                 (+ shear terms             MHDRunGodunov.cpp:2448-2525)
//   mhd_trace3d_cell   U,Q,E        -> T   (38)   slopes + CTU trace         ..._cpu_v3.cpp:115-361, trace_mhd.h:1854-2248
//   mhd_flux3d_cell    T            -> F   (15), emf (3)   3 HLLD + 3 2D-HLLD ..._cpu_v3.cpp:372-583
//   mhd_update3d_column Uold,F,emf  -> Unew (8)   conservative + CT update   ..._cpu_v3.cpp:475-533,600-630
//
// T is the COMPACT traced state: instead of the reference's 18 face/edge state arrays (144 doubles per cell,
// trace_mhd.h:2032-2246) we keep what they are all built from -- the time-advanced cell state, the three
// advanced LOW-face field components and the limited half slopes (38 doubles) -- and rebuild qm/qp/qEdge on the
// fly in the flux kernel with exactly the reference's additions.  The HIGH-face values of a cell are the LOW-face
// values of its +1 neighbour (bit-identical expressions), so they are read from there.
#pragma once
#include "dev_numerics.h"

namespace rgpu_dev {

struct IJK { int i, j, k; };
RG_DEVFN IJK unflatten(const DevParams& g, unsigned idx) {
  IJK c;
  c.i = (int)(idx % (unsigned)g.isize);
  const unsigned t = idx / (unsigned)g.isize;
  c.j = (int)(t % (unsigned)g.jsize);
  c.k = (int)(t / (unsigned)g.jsize);
  return c;
}

// slots of the compact traced state T
enum {
  T_R = 0, T_P, T_U, T_V, T_W, T_A, T_B, T_C,       // advanced cell-centred state
  T_AL, T_BL, T_CL,                                  // advanced low-face field
  T_DX,                                              // 11..17: x half slopes of r,p,u,v,w then B, C
  T_DY = T_DX + 7,                                   // 18..24: y half slopes of r,p,u,v,w then A, C
  T_DZ = T_DY + 7,                                   // 25..31: z half slopes of r,p,u,v,w then A, B
  T_DALY = T_DZ + 7, T_DALZ, T_DBLX, T_DBLZ, T_DCLX, T_DCLY,   // 32..37: transverse half slopes of the low faces
  T_COUNT                                            // 38
};
// flux array F: 5 hydro components per direction, in the FACE-NORMAL frame
enum { F_X = 0, F_Y = 5, F_Z = 10, F_COUNT = 15 };
enum { EMF_Z = 0, EMF_Y = 1, EMF_X = 2 };            // EmfIndex, constants.h:191-195

// Accessors of the compact traced state.  The flat kernels keep T as a global SoA array (component stride ncell, cell
// index = flat array index); the LDS-tiled sweep of the HIP backend (hip/tiled_mhd.h) keeps two planes of an x-y
// tile in LDS.  The trace / face-state / edge-state code below is written once against this interface:
//   get(slot, m)   component `slot` of cell m            stride(D)   distance of the +1 neighbour along D
//   writer.put(slot, value)                              (m and strides are whatever the accessor's index space is)
struct TGlobalRead {
  const double* T; size_t N; unsigned sj, sk;
  RG_DEVFN double get(int slot, unsigned m) const { return T[m + (size_t)slot * N]; }
  RG_DEVFN unsigned stride(int D) const { return (D == XD) ? 1u : (D == YD) ? sj : sk; }
};
// Inputs of the trace around a cell: primitives q, face-centred field bf (U components IA, IB, IC) and edge electric
// field e, addressed by an in-plane cell index m (neighbours: m +- 1, m +- sj()) and a plane offset dz in {-1, 0, +1}.
struct TraceInGlobal {
  const double* U; const double* Q; const double* E; size_t N; unsigned sj_, sk_;
  RG_DEVFN double q(int v, int dz, unsigned m) const { return Q[(size_t)v * N + (m + (unsigned)(dz * (int)sk_))]; }
  RG_DEVFN double bf(int comp, int dz, unsigned m) const { return U[(size_t)(IA + comp) * N + (m + (unsigned)(dz * (int)sk_))]; }
  RG_DEVFN double e(int comp, int dz, unsigned m) const { return E[(size_t)comp * N + (m + (unsigned)(dz * (int)sk_))]; }
  RG_DEVFN unsigned sj() const { return sj_; }
  RG_DEVFN void e_ready() const {}   // called once before the first e(): the tiled kernel waits for its E producers here
};
struct TGlobalWrite {   // bound to one cell
  double* t; size_t N;
  RG_DEVFN void put(int slot, double v) const { RG_STREAM_STORE(&t[(size_t)slot * N], v); }
};

// ------------------------------------------------------------------------------------------------------------
// primitive variables (2D and 3D)
// ------------------------------------------------------------------------------------------------------------
RG_DEVFN void mhd_prim_cell(const DevParams& g, const double* __restrict__ U, double* __restrict__ Q, double dt, unsigned idx) {
  const IJK c = unflatten(g, idx);
  if (c.i >= g.isize - 1 || c.j >= g.jsize - 1) return;
  if (g.three_d && c.k >= g.ksize - 1) return;
  const size_t N = g.ncell;
  double u[8];
#pragma unroll
  for (int v = 0; v < 8; ++v) u[v] = U[idx + v * N];
  const double bnx = U[idx + 1 + IA * N];
  const double bny = U[idx + g.sj + IB * N];
  const double bnz = g.three_d ? U[idx + g.sk + IC * N] : 0.0;
  const Prim8 q = mhd_prim(g, u, bnx, bny, bnz, dt);
  Q[idx + ID * N] = q.r; Q[idx + IP * N] = q.p; Q[idx + IU * N] = q.u; Q[idx + IV * N] = q.v;
  Q[idx + IW * N] = q.w; Q[idx + IA * N] = q.a; Q[idx + IB * N] = q.b; Q[idx + IC * N] = q.c;
}

// ------------------------------------------------------------------------------------------------------------
// edge-centred electric field, defined at the low corner edges of cell (i,j,k)
// ------------------------------------------------------------------------------------------------------------
// One component (0 = Ex, 1 = Ey, 2 = Ez) of the electric field at the low corner edges of cell m, from the primitives /
// face field around it (accessor interface of the trace inputs; planes dz = -1, 0).  xPos = cell-centre x of cell m.
template <int COMP, class TIN>
RG_DEVFN double mhd_elec_comp(const DevParams& g, const TIN& in, double xPos, unsigned m) {
  const unsigned sj = in.sj();
  double u, v, w, A, B, C, e;
  if (COMP == 0) {   // Ex : average over the 4 cells around the x-edge (j-1..j, k-1..k)
    v = 0.25 * (in.q(IV, -1, m - sj) + in.q(IV, 0, m - sj) + in.q(IV, -1, m) + in.q(IV, 0, m));
    w = 0.25 * (in.q(IW, -1, m - sj) + in.q(IW, 0, m - sj) + in.q(IW, -1, m) + in.q(IW, 0, m));
    B = 0.5 * (in.bf(1, -1, m) + in.bf(1, 0, m));
    C = 0.5 * (in.bf(2, 0, m - sj) + in.bf(2, 0, m));
    e = v * C - w * B;
    if (g.rot) { const double shear = -1.5 * g.Omega0 * xPos; e += shear * C; }
    return e;
  }
  if (COMP == 1) {   // Ey
    u = 0.25 * (in.q(IU, -1, m - 1) + in.q(IU, 0, m - 1) + in.q(IU, -1, m) + in.q(IU, 0, m));
    w = 0.25 * (in.q(IW, -1, m - 1) + in.q(IW, 0, m - 1) + in.q(IW, -1, m) + in.q(IW, 0, m));
    A = 0.5 * (in.bf(0, -1, m) + in.bf(0, 0, m));
    C = 0.5 * (in.bf(2, 0, m - 1) + in.bf(2, 0, m));
    return w * A - u * C;
  }
  // Ez
  u = 0.25 * (in.q(IU, 0, m - 1 - sj) + in.q(IU, 0, m - 1) + in.q(IU, 0, m - sj) + in.q(IU, 0, m));
  v = 0.25 * (in.q(IV, 0, m - 1 - sj) + in.q(IV, 0, m - 1) + in.q(IV, 0, m - sj) + in.q(IV, 0, m));
  A = 0.5 * (in.bf(0, 0, m - sj) + in.bf(0, 0, m));
  B = 0.5 * (in.bf(1, 0, m - 1) + in.bf(1, 0, m));
  e = u * B - v * A;
  if (g.rot) { const double shear = -1.5 * g.Omega0 * (xPos - g.dx / 2); e -= shear * A; }
  return e;
}

RG_DEVFN bool elec_in_range(const DevParams& g, const IJK c) {
  return !(c.i < 1 || c.i >= g.isize - 1 || c.j < 1 || c.j >= g.jsize - 1 || c.k < 1 || c.k >= g.ksize - 1);
}

RG_DEVFN void mhd_elec_cell(const DevParams& g, const double* __restrict__ U, const double* __restrict__ Q,
                            double* __restrict__ E, unsigned idx) {
  const IJK c = unflatten(g, idx);
  if (!elec_in_range(g, c)) return;
  const size_t N = g.ncell;
  const double xPos = g.xMin + g.dx / 2 + (c.i - g.gw) * g.dx;
  const TraceInGlobal in = {U, Q, 0, N, g.sj, g.sk};
  E[idx] = mhd_elec_comp<0>(g, in, xPos, idx);
  E[idx + N] = mhd_elec_comp<1>(g, in, xPos, idx);
  E[idx + 2 * N] = mhd_elec_comp<2>(g, in, xPos, idx);
}

// ------------------------------------------------------------------------------------------------------------
// slopes + MUSCL-Hancock / CTU trace -> compact state
// ------------------------------------------------------------------------------------------------------------
// Second half of the trace: from the cell state q, its limited full slopes in
// x,y,z and the 9 edge electric fields E9 = {Ex(j,k),Ex(j,k+1),Ex(j+1,k), Ey(i,k),Ey(i,k+1),Ey(i+1,k),
// Ez(i,j),Ez(i,j+1),Ez(i+1,j)} to the compact traced state.
template <class TIN, class TW>
RG_DEVFN void mhd_trace3d_finish(const DevParams& g, const TIN& in, const TW& tw, const IJK c,
                                 unsigned m, const double* q, const double* dx_, const double* dy_, const double* dz_,
                                 const double* E9, double dtdx, double dtdy, double dtdz) {
  const unsigned sj = in.sj();
  double AL = in.bf(0, 0, m), BL = in.bf(1, 0, m), CL = in.bf(2, 0, m);
  const double AR = in.bf(0, 0, m + 1), BR = in.bf(1, 0, m + sj), CR = in.bf(2, +1, m);
  // transverse slopes of the low-face field (slope_unsplit_mhd_3d: slope type capped at 2)
  const double mst = g.mag_slope_type;
  const double dALy = tvd_half_slope(mst, in.bf(0, 0, m - sj), AL, in.bf(0, 0, m + sj));
  const double dALz = tvd_half_slope(mst, in.bf(0, -1, m), AL, in.bf(0, +1, m));
  const double dBLx = tvd_half_slope(mst, in.bf(1, 0, m - 1), BL, in.bf(1, 0, m + 1));
  const double dBLz = tvd_half_slope(mst, in.bf(1, -1, m), BL, in.bf(1, +1, m));
  const double dCLx = tvd_half_slope(mst, in.bf(2, 0, m - 1), CL, in.bf(2, 0, m + 1));
  const double dCLy = tvd_half_slope(mst, in.bf(2, 0, m - sj), CL, in.bf(2, 0, m + sj));

  // electric field at the edges bounding the three low faces
  const double ELL = E9[0], ELR = E9[1], ERL = E9[2];
  const double FLL = E9[3], FLR = E9[4], FRL = E9[5];
  const double GLL = E9[6], GLR = E9[7], GRL = E9[8];

  double r = q[ID], p = q[IP], u = q[IU], v = q[IV], w = q[IW], A = q[IA], B = q[IB], C = q[IC];
  // (dx_, dy_, dz_ arrive as HALF slopes: the reference's "0.5 * dq" is taken where the slope is formed)
  const double drx = dx_[ID], dpx = dx_[IP], dux = dx_[IU], dvx = dx_[IV], dwx = dx_[IW], dCx = dx_[IC], dBx = dx_[IB];
  const double dry = dy_[ID], dpy = dy_[IP], duy = dy_[IU], dvy = dy_[IV], dwy = dy_[IW], dCy = dy_[IC], dAy = dy_[IA];
  const double drz = dz_[ID], dpz = dz_[IP], duz = dz_[IU], dvz = dz_[IV], dwz = dz_[IW], dAz = dz_[IA], dBz = dz_[IB];
  const double dAx = 0.5 * (AR - AL), dBy = 0.5 * (BR - BL), dCz = 0.5 * (CR - CL);
  const double gamma = g.gamma0;

  double sr0 = (-u * drx - dux * r) * dtdx + (-v * dry - dvy * r) * dtdy + (-w * drz - dwz * r) * dtdz;
  const rg_recip_t inv_r = rg_recip(r);   // nine divisions by the density
  double su0 = (-u * dux - rg_div(dpx + B * dBx + C * dCx, inv_r)) * dtdx + (-v * duy + rg_div(B * dAy, inv_r)) * dtdy + (-w * duz + rg_div(C * dAz, inv_r)) * dtdz;
  double sv0 = (-u * dvx + rg_div(A * dBx, inv_r)) * dtdx + (-v * dvy - rg_div(dpy + A * dAy + C * dCy, inv_r)) * dtdy + (-w * dvz + rg_div(C * dBz, inv_r)) * dtdz;
  double sw0 = (-u * dwx + rg_div(A * dCx, inv_r)) * dtdx + (-v * dwy + rg_div(B * dCy, inv_r)) * dtdy + (-w * dwz - rg_div(dpz + A * dAz + B * dBz, inv_r)) * dtdz;
  double sp0 = (-u * dpx - dux * gamma * p) * dtdx + (-v * dpy - dvy * gamma * p) * dtdy + (-w * dpz - dwz * gamma * p) * dtdz;
  double sA0 = (u * dBy + B * duy - v * dAy - A * dvy) * dtdy + (u * dCz + C * duz - w * dAz - A * dwz) * dtdz;
  double sB0 = (v * dAx + A * dvx - u * dBx - B * dux) * dtdx + (v * dCz + C * dvz - w * dBz - B * dwz) * dtdz;
  double sC0 = (w * dAx + A * dwx - u * dCx - C * dux) * dtdx + (w * dBy + B * dwy - v * dCy - C * dvy) * dtdy;
  if (g.Omega0 > 0) {
    const double xPos = g.xMin + g.dx / 2 + (c.i - g.gw) * g.dx;
    const double shear = -1.5 * g.Omega0 * xPos;
    sr0 = sr0 - shear * dry * dtdy;
    su0 = su0 - shear * duy * dtdy;
    sv0 = sv0 - shear * dvy * dtdy;
    sw0 = sw0 - shear * dwy * dtdy;
    sp0 = sp0 - shear * dpy * dtdy;
    sA0 = sA0 - shear * dAy * dtdy;
    sB0 = sB0 + (shear * dAx - 1.5 * g.Omega0 * A * g.dx) * dtdx + shear * dBz * dtdz;
    sC0 = sC0 - shear * dCy * dtdy;
  }
  const double sAL0 = +(GLR - GLL) * dtdy * 0.5 - (FLR - FLL) * dtdz * 0.5;
  const double sBL0 = -(GRL - GLL) * dtdx * 0.5 + (ELR - ELL) * dtdz * 0.5;
  const double sCL0 = +(FRL - FLL) * dtdx * 0.5 - (ERL - ELL) * dtdy * 0.5;

  r = r + sr0; u = u + su0; v = v + sv0; w = w + sw0; p = p + sp0; A = A + sA0; B = B + sB0; C = C + sC0;
  AL = AL + sAL0; BL = BL + sBL0; CL = CL + sCL0;

  tw.put(T_R, r); tw.put(T_P, p); tw.put(T_U, u); tw.put(T_V, v); tw.put(T_W, w); tw.put(T_A, A); tw.put(T_B, B); tw.put(T_C, C);
  tw.put(T_AL, AL); tw.put(T_BL, BL); tw.put(T_CL, CL);
  tw.put(T_DX + 0, drx); tw.put(T_DX + 1, dpx); tw.put(T_DX + 2, dux); tw.put(T_DX + 3, dvx); tw.put(T_DX + 4, dwx);
  tw.put(T_DX + 5, dBx); tw.put(T_DX + 6, dCx);
  tw.put(T_DY + 0, dry); tw.put(T_DY + 1, dpy); tw.put(T_DY + 2, duy); tw.put(T_DY + 3, dvy); tw.put(T_DY + 4, dwy);
  tw.put(T_DY + 5, dAy); tw.put(T_DY + 6, dCy);
  tw.put(T_DZ + 0, drz); tw.put(T_DZ + 1, dpz); tw.put(T_DZ + 2, duz); tw.put(T_DZ + 3, dvz); tw.put(T_DZ + 4, dwz);
  tw.put(T_DZ + 5, dAz); tw.put(T_DZ + 6, dBz);
  tw.put(T_DALY, dALy); tw.put(T_DALZ, dALz); tw.put(T_DBLX, dBLx); tw.put(T_DBLZ, dBLz); tw.put(T_DCLX, dCLx); tw.put(T_DCLY, dCLy);
}

RG_DEVFN bool trace3d_in_range(const DevParams& g, const IJK c) {
  const int lo = g.gw - 1;
  return !(c.i < lo || c.i > g.isize - g.gw || c.j < lo || c.j > g.jsize - g.gw || c.k < lo || c.k > g.ksize - g.gw);
}

// Front end: Q and E come from the arrays written by mhd_prim_cell and mhd_elec_cell.  (A variant that recomputes
// them from U inside this kernel -- no Q/E round trip, no prim/elec launches -- was measured NOT faster: 240 VGPRs,
// 77.6-78.3 vs 76.6-77.3 ms/step at 512^3; it was dropped.)
// mhd_trace3d_at: cell c (in-plane index m of the input accessor) is known to be inside the trace range; tw receives
// the 38 components
template <class TIN, class TW>
RG_DEVFN void mhd_trace3d_at(const DevParams& g, const TIN& in, const TW& tw, double dtdx, double dtdy,
                             double dtdz, const IJK c, unsigned m) {
  const unsigned sj = in.sj();
  const double st = g.slope_type;
  double q[8], dx_[8], dy_[8], dz_[8];
#pragma unroll
  for (int v = 0; v < 8; ++v) {
    q[v] = in.q(v, 0, m);
    if (st == 0) { dx_[v] = 0.0; dy_[v] = 0.0; dz_[v] = 0.0; }
    else if (st == 3) {   // plain path only (mhd_godunov_unsplit_cpu_v3.cpp:196-211); rejected at create for Omega0 > 0
      double lo = q[v], hi = q[v];
#pragma unroll
      for (int dk = -1; dk <= 1; ++dk)
        for (int dj = -1; dj <= 1; ++dj)
#pragma unroll
          for (int di = -1; di <= 1; ++di) {
            const double nb = in.q(v, dk, (unsigned)((int)m + di + dj * (int)sj));
            lo = (nb < lo) ? nb : lo;
            hi = (nb > hi) ? nb : hi;
          }
      const double dfx = 0.5 * (in.q(v, 0, m + 1) - in.q(v, 0, m - 1)), dfy = 0.5 * (in.q(v, 0, m + sj) - in.q(v, 0, m - sj));
      const double dfz = 0.5 * (in.q(v, +1, m) - in.q(v, -1, m));
      const double dlim = positivity_limiter(lo, hi, q[v], fabs(dfx) + fabs(dfy) + fabs(dfz));
      dx_[v] = dlim * dfx * 0.5;
      dy_[v] = dlim * dfy * 0.5;
      dz_[v] = dlim * dfz * 0.5;
    } else {
      dx_[v] = tvd_half_slope(st, in.q(v, 0, m - 1), q[v], in.q(v, 0, m + 1));
      dy_[v] = tvd_half_slope(st, in.q(v, 0, m - sj), q[v], in.q(v, 0, m + sj));
      dz_[v] = tvd_half_slope(st, in.q(v, -1, m), q[v], in.q(v, +1, m));
    }
  }
  in.e_ready();
  const double E9[9] = {in.e(0, 0, m), in.e(0, +1, m), in.e(0, 0, m + sj), in.e(1, 0, m), in.e(1, +1, m), in.e(1, 0, m + 1),
                        in.e(2, 0, m), in.e(2, 0, m + sj), in.e(2, 0, m + 1)};
  mhd_trace3d_finish(g, in, tw, c, m, q, dx_, dy_, dz_, E9, dtdx, dtdy, dtdz);
}

RG_DEVFN void mhd_trace3d_cell(const DevParams& g, const double* __restrict__ U, const double* __restrict__ Q,
                               const double* __restrict__ E, double* __restrict__ T, double dtdx, double dtdy,
                               double dtdz, unsigned idx) {
  const IJK c = unflatten(g, idx);
  if (!trace3d_in_range(g, c)) return;
  const TGlobalWrite tw = {T + idx, (size_t)g.ncell};
  const TraceInGlobal in = {U, Q, E, (size_t)g.ncell, g.sj, g.sk};
  mhd_trace3d_at(g, in, tw, dtdx, dtdy, dtdz, c, idx);
}

// ------------------------------------------------------------------------------------------------------------
// rebuilding face and edge states from T
// ------------------------------------------------------------------------------------------------------------

// 3D trace floors: rho >= smallr, p >= smallp (no density factor, trace_mhd.h:2041-2042)
RG_DEVFN void floor3d(const DevParams& g, Prim8& s) {
  s.r = fmax(g.smallr, s.r);
  s.p = fmax(g.smallp, s.p);
}

// Face state of cell m in direction D, in the face-NORMAL frame.  SIDE=+1: the reference's qm[D] (state at the
// HIGH face of m, the LEFT state of face m+1); SIDE=-1: qp[D] (state at the LOW face of m, the RIGHT state).
// gm = GLOBAL flat index of cell m (only the per-cell gravity field reads it)
template <int D, int SIDE, bool GF, class TA>
RG_DEVFN Prim8 face_state3d(const DevParams& g, const TA& T, unsigned m, unsigned gm) {
  const unsigned sD = T.stride(D);
  const int S = (D == XD) ? T_DX : (D == YD) ? T_DY : T_DZ;
  const double s = (double)SIDE;
  double r = T.get(T_R, m) + s * T.get(S + 0, m);
  double p = T.get(T_P, m) + s * T.get(S + 1, m);
  double u = T.get(T_U, m) + s * T.get(S + 2, m);
  double v = T.get(T_V, m) + s * T.get(S + 3, m);
  double w = T.get(T_W, m) + s * T.get(S + 4, m);
  if (GF || g.grav_on) {   // gravity predictor (..._cpu_v3.cpp:277-290)
    double gx, gy, gz;
    half_dt_gravity<GF>(g, gm, gx, gy, gz);
    u += gx; v += gy; w += gz;
  }
  // normal field: the advanced face value (own low face, or the +1 neighbour's low face for the high side)
  const int TF = (D == XD) ? T_AL : (D == YD) ? T_BL : T_CL;
  const double bn = (SIDE > 0) ? T.get(TF, m + sD) : T.get(TF, m);
  // the two transverse cell-centred components with their slopes along D (slots 5,6 of the slope group)
  double b1, b2;  // in grid order: the two components other than D, ascending
  if (D == XD) { b1 = T.get(T_B, m) + s * T.get(S + 5, m); b2 = T.get(T_C, m) + s * T.get(S + 6, m); }
  else if (D == YD) { b1 = T.get(T_A, m) + s * T.get(S + 5, m); b2 = T.get(T_C, m) + s * T.get(S + 6, m); }
  else { b1 = T.get(T_A, m) + s * T.get(S + 5, m); b2 = T.get(T_B, m) + s * T.get(S + 6, m); }
  Prim8 o;
  o.r = r; o.p = p;
  // permutation into the normal frame: x: (u,v,w | A,B,C)  y: (v,u,w | B,A,C)  z: (w,v,u | C,B,A)
  if (D == XD) { o.u = u; o.v = v; o.w = w; o.a = bn; o.b = b1; o.c = b2; }
  else if (D == YD) { o.u = v; o.v = u; o.w = w; o.a = bn; o.b = b1; o.c = b2; }
  else { o.u = w; o.v = v; o.w = u; o.a = bn; o.b = b2; o.c = b1; }
  floor3d(g, o);
  return o;
}

// Edge state of cell m for the edge along direction EDIR (0=x,1=y,2=z), at the corner given by the signs
// (S1,S2) along the two transverse directions (t1,t2) = (y,z) | (z,x) | (x,y), returned in the EDGE frame
// (u,v,w / a,b,c = components along t1, t2, e).  Reproduces qEdge of trace_mhd.h:2104-2246.
template <int EDIR, int S1, int S2, bool GF, class TA>
RG_DEVFN Prim8 edge_state3d(const DevParams& g, const TA& T, unsigned m, unsigned gm) {
  const int t1 = (EDIR + 1) % 3, t2 = (EDIR + 2) % 3;
  const unsigned st1 = T.stride(t1);
  const unsigned st2 = T.stride(t2);
  const int G1 = (t1 == XD) ? T_DX : (t1 == YD) ? T_DY : T_DZ;
  const int G2 = (t2 == XD) ? T_DX : (t2 == YD) ? T_DY : T_DZ;
  const double s1 = (double)S1, s2 = (double)S2;
  // cell-centred quantities: q + (s1*d_t1 + s2*d_t2).  In the reference the x-direction slope always comes
  // first inside the parenthesis, then y, then z; keep that operand order.
  const bool t1_first = t1 < t2;
#define RG_EDGE_SUM(base, k1, k2) \
  (T.get((base), m) + (t1_first ? (s1 * T.get(G1 + (k1), m) + s2 * T.get(G2 + (k2), m)) : (s2 * T.get(G2 + (k2), m) + s1 * T.get(G1 + (k1), m))))
  double vel[3];
  Prim8 o;
  o.r = RG_EDGE_SUM(T_R, 0, 0);
  o.p = RG_EDGE_SUM(T_P, 1, 1);
  vel[0] = RG_EDGE_SUM(T_U, 2, 2);
  vel[1] = RG_EDGE_SUM(T_V, 3, 3);
  vel[2] = RG_EDGE_SUM(T_W, 4, 4);
  if (GF || g.grav_on) {   // gravity predictor (..._cpu_v3.cpp:292-330)
    double gx, gy, gz;
    half_dt_gravity<GF>(g, gm, gx, gy, gz);
    vel[0] += gx; vel[1] += gy; vel[2] += gz;
  }
  // the field component along the edge is cell centred; its slope slot inside a direction group:
  //   group X holds (B,C) at 5,6 ; group Y holds (A,C) at 5,6 ; group Z holds (A,B) at 5,6
  const int Te = (EDIR == XD) ? T_A : (EDIR == YD) ? T_B : T_C;
  const int k_in_G1 = (t1 == XD) ? ((EDIR == YD) ? 5 : 6) : (t1 == YD) ? ((EDIR == XD) ? 5 : 6) : ((EDIR == XD) ? 5 : 6);
  const int k_in_G2 = (t2 == XD) ? ((EDIR == YD) ? 5 : 6) : (t2 == YD) ? ((EDIR == XD) ? 5 : 6) : ((EDIR == XD) ? 5 : 6);
  const double be = RG_EDGE_SUM(Te, k_in_G1, k_in_G2);
#undef RG_EDGE_SUM
  // the two in-plane components are face centred: take the face on the signed side and add the signed
  // transverse half slope of THAT face (stored with the cell owning the face)
  const int TF1 = (t1 == XD) ? T_AL : (t1 == YD) ? T_BL : T_CL;   // face normal to t1
  const int TF2 = (t2 == XD) ? T_AL : (t2 == YD) ? T_BL : T_CL;   // face normal to t2
  // slope of face-t1 field along t2 / of face-t2 field along t1
  const int TS1 = (t1 == XD) ? ((t2 == YD) ? T_DALY : T_DALZ) : (t1 == YD) ? ((t2 == XD) ? T_DBLX : T_DBLZ) : ((t2 == XD) ? T_DCLX : T_DCLY);
  const int TS2 = (t2 == XD) ? ((t1 == YD) ? T_DALY : T_DALZ) : (t2 == YD) ? ((t1 == XD) ? T_DBLX : T_DBLZ) : ((t1 == XD) ? T_DCLX : T_DCLY);
  const unsigned m1 = (S1 > 0) ? m + st1 : m;
  const unsigned m2 = (S2 > 0) ? m + st2 : m;
  const double b1 = T.get(TF1, m1) + s2 * T.get(TS1, m1);
  const double b2 = T.get(TF2, m2) + s1 * T.get(TS2, m2);
  o.u = vel[t1]; o.v = vel[t2]; o.w = vel[EDIR];
  o.a = b1; o.b = b2; o.c = be;
  floor3d(g, o);
  return o;
}

// ------------------------------------------------------------------------------------------------------------
// Riemann problems at the three low faces and the three low edges of cell (i,j,k)
// ------------------------------------------------------------------------------------------------------------
enum { DO_FLUX_X = 1, DO_FLUX_Y = 2, DO_FLUX_Z = 4, DO_EMF_X = 8, DO_EMF_Y = 16, DO_EMF_Z = 32, DO_ALL = 63 };

template <int D>
RG_DEVFN void store_flux(double* __restrict__ F, size_t N, unsigned idx, const double* fl) {   // idx: flux_index of the cell, N = g.fN
  const int base = (D == XD) ? F_X : (D == YD) ? F_Y : F_Z;
#pragma unroll
  for (int v = 0; v < 5; ++v) RG_STREAM_STORE(&F[idx + (size_t)(base + v) * N], fl[v]);
}

// 1D Riemann problem at a face normal to D between the states L, R (face-normal frame) -> fl[0..4] used by the callers;
// in the rotating frame the y flux gets the shear advection of the upwind state as left by the Riemann solver
// (MHDRunGodunov.cpp:2861-2899)
template <int D>
RG_DEVFN void mhd_face_flux(const DevParams& g, Prim8& L, Prim8& R, double xPos, double* fl) {
#pragma unroll
  for (int v = 0; v < 8; ++v) fl[v] = 0.0;
  mhd_riemann(g, L, R, fl);
  if (D == YD && g.rot) {
    const double shear_y = -1.5 * g.Omega0 * xPos;
    const double bn_mean = 0.5 * (L.a + R.a);
    const Prim8& s = (shear_y > 0) ? L : R;
    const double eMag = 0.5 * (s.a * s.a + s.b * s.b + s.c * s.c);
    const double eKin = 0.5 * (s.u * s.u + s.v * s.v + s.w * s.w);
    const double eTot = eKin + eMag + s.p / (g.gamma0 - 1.0);
    fl[ID] = fl[ID] + shear_y * s.r;
    fl[IP] = fl[IP] + shear_y * (eTot + eMag - bn_mean * bn_mean);
    fl[IU] = fl[IU] + shear_y * s.r * s.u;
    fl[IV] = fl[IV] + shear_y * s.r * s.v;
    fl[IW] = fl[IW] + shear_y * s.r * s.w;
  }
}

// The Riemann problems selected by MASK at the low faces / low edges of the cell whose traced state is T(m); idx is its
// global flat index (gravity field), fidx its flux_index (output location), xPos its x coordinate.  The cell is known to be in range.
template <int MASK, bool GF, class TA>
RG_DEVFN void mhd_flux3d_at(const DevParams& g, const TA& T, unsigned m, double xPos, double* __restrict__ F,
                            double* __restrict__ emf, unsigned idx, unsigned fidx) {
  const size_t N = g.fN;
  const unsigned sx = T.stride(XD), sj = T.stride(YD), sk = T.stride(ZD);
  const unsigned gsj = g.sj, gsk = g.sk;
  double fl[8];
  if (MASK & DO_FLUX_X) {
    Prim8 L = face_state3d<XD, +1, GF>(g, T, m - sx, idx - 1), R = face_state3d<XD, -1, GF>(g, T, m, idx);
    mhd_face_flux<XD>(g, L, R, xPos, fl);
    store_flux<XD>(F, N, fidx, fl);
  }
  if (MASK & DO_FLUX_Y) {
    Prim8 L = face_state3d<YD, +1, GF>(g, T, m - sj, idx - gsj), R = face_state3d<YD, -1, GF>(g, T, m, idx);
    mhd_face_flux<YD>(g, L, R, xPos, fl);
    store_flux<YD>(F, N, fidx, fl);
  }
  if (MASK & DO_FLUX_Z) {
    Prim8 L = face_state3d<ZD, +1, GF>(g, T, m - sk, idx - gsk), R = face_state3d<ZD, -1, GF>(g, T, m, idx);
    mhd_face_flux<ZD>(g, L, R, xPos, fl);
    store_flux<ZD>(F, N, fidx, fl);
  }
  // EMFs: slot order (RT, RB, LT, LB) = (+,+) from c-t1-t2, (+,-) from c-t1, (-,+) from c-t2, (-,-) from c
  if (MASK & DO_EMF_Z) {  // t1 = x, t2 = y
    const Prim8 rt = edge_state3d<2, +1, +1, GF>(g, T, m - sx - sj, idx - 1 - gsj), rb = edge_state3d<2, +1, -1, GF>(g, T, m - sx, idx - 1);
    const Prim8 lt = edge_state3d<2, -1, +1, GF>(g, T, m - sj, idx - gsj), lb = edge_state3d<2, -1, -1, GF>(g, T, m, idx);
    RG_STREAM_STORE(&emf[fidx + (size_t)EMF_Z * N], edge_emf<2>(g, rt, rb, lt, lb, xPos));
  }
  if (MASK & DO_EMF_Y) {  // t1 = z, t2 = x
    const Prim8 rt = edge_state3d<1, +1, +1, GF>(g, T, m - sk - sx, idx - gsk - 1), rb = edge_state3d<1, +1, -1, GF>(g, T, m - sk, idx - gsk);
    const Prim8 lt = edge_state3d<1, -1, +1, GF>(g, T, m - sx, idx - 1), lb = edge_state3d<1, -1, -1, GF>(g, T, m, idx);
    RG_STREAM_STORE(&emf[fidx + (size_t)EMF_Y * N], edge_emf<1>(g, rt, rb, lt, lb, xPos));
  }
  if (MASK & DO_EMF_X) {  // t1 = y, t2 = z
    const Prim8 rt = edge_state3d<0, +1, +1, GF>(g, T, m - sj - sk, idx - gsj - gsk), rb = edge_state3d<0, +1, -1, GF>(g, T, m - sj, idx - gsj);
    const Prim8 lt = edge_state3d<0, -1, +1, GF>(g, T, m - sk, idx - gsk), lb = edge_state3d<0, -1, -1, GF>(g, T, m, idx);
    RG_STREAM_STORE(&emf[fidx + (size_t)EMF_X * N], edge_emf<0>(g, rt, rb, lt, lb, xPos));
  }
}

RG_DEVFN bool flux3d_in_range(const DevParams& g, const IJK c) {
  return !(c.i < g.gw || c.i > g.isize - g.gw || c.j < g.gw || c.j > g.jsize - g.gw || c.k < g.gw || c.k > g.ksize - g.gw);
}

template <int MASK, bool GF>
RG_DEVFN void mhd_flux3d_cell(const DevParams& g, const double* __restrict__ T, double* __restrict__ F,
                              double* __restrict__ emf, unsigned idx) {
  const IJK c = unflatten(g, idx);
  if (!flux3d_in_range(g, c)) return;
  const double xPos = g.xMin + g.dx / 2 + (c.i - g.gw) * g.dx;
  const TGlobalRead ta = {T, (size_t)g.ncell, g.sj, g.sk};
  mhd_flux3d_at<MASK, GF>(g, ta, idx, xPos, F, emf, idx, flux_index(g, c.i, c.j, c.k));
}

// ------------------------------------------------------------------------------------------------------------
// shearing box: remap of the x-border density flux and of emfY (MHDRunGodunov.cpp:3203-3300)
// ------------------------------------------------------------------------------------------------------------
struct ShearRemap {   // computed on the host from totalTime, dt (fmod / integer part), MHDRunGodunov.cpp:3213-3216
  int jplus;
  double eps_min;     // 1 - epsi/dy  (weight used for the inner / xmin border)
  double eps_max;     // epsi/dy      (outer / xmax border)
};

// step 1: save the two emfY columns (they are rewritten in place by step 2).  idx2 = j + jsize*k
RG_DEVFN void shear_save_emf_cell(const DevParams& g, const double* __restrict__ emf, double* __restrict__ save, unsigned idx2) {
  const unsigned j = idx2 % (unsigned)g.jsize, k = idx2 / (unsigned)g.jsize;
  const size_t N = g.fN;
  const size_t row = (size_t)g.fsj * j + (size_t)g.fsk * k + g.foff;
  const size_t P = (size_t)g.jsize * g.ksize;
  save[idx2] = emf[row + g.gw + (size_t)EMF_Y * N];
  save[idx2 + P] = emf[row + g.nx + g.gw + (size_t)EMF_Y * N];
}

// step 2: remapped density flux (into remap[0..P) for xmin, [P..2P) for xmax) and averaged emfY at both borders
RG_DEVFN void shear_remap_cell(const DevParams& g, const ShearRemap sr, const double* __restrict__ F,
                               double* __restrict__ emf, const double* __restrict__ save, double* __restrict__ remap,
                               double dtdx, unsigned idx2) {
  const int j = (int)(idx2 % (unsigned)g.jsize), k = (int)(idx2 / (unsigned)g.jsize);
  const size_t N = g.fN;
  const size_t P = (size_t)g.jsize * g.ksize;
  const int gw = g.gw, ny = g.ny, nx = g.nx;
  const double* Fd = F + (size_t)(F_X + ID) * N;   // density flux through the low x face
  const size_t krow = (size_t)g.fsk * k + g.foff;   // (F / emf have their own pitch: flux_index)
  const size_t fsj = g.fsj;
  const bool inner = (j >= gw && j < g.jsize - gw + 1 && k >= gw && k < g.ksize - gw + 1);
  // ---- xmin border: looks at the xmax border, shifted by -(jplus+1) ----
  {
    int jremap = j - sr.jplus - 1, jremapp1 = jremap + 1;
    const double eps = sr.eps_min;
    if (jremap < gw) jremap += ny;
    if (jremapp1 < gw) jremapp1 += ny;
    if (inner) {
      const double own = Fd[krow + fsj * j + gw] * dtdx;
      const double o0 = Fd[krow + fsj * jremap + nx + gw] * dtdx;
      const double o1 = Fd[krow + fsj * jremapp1 + nx + gw] * dtdx;
      double rv = own + (1.0 - eps) * o0 + eps * o1;
      rv *= 0.5;
      remap[idx2] = rv;
    }
    double e = save[idx2];
    e += (1.0 - eps) * save[P + jremap + (size_t)g.jsize * k] + eps * save[P + jremapp1 + (size_t)g.jsize * k];
    e *= 0.5;
    emf[krow + fsj * j + gw + (size_t)EMF_Y * N] = e;
  }
  // ---- xmax border: looks at the xmin border, shifted by +jplus ----
  {
    int jremap = j + sr.jplus, jremapp1 = jremap + 1;
    const double eps = sr.eps_max;
    if (jremap > ny + gw - 1) jremap -= ny;
    if (jremapp1 > ny + gw - 1) jremapp1 -= ny;
    if (inner) {
      const double own = Fd[krow + fsj * j + nx + gw] * dtdx;
      const double o0 = Fd[krow + fsj * jremap + gw] * dtdx;
      const double o1 = Fd[krow + fsj * jremapp1 + gw] * dtdx;
      double rv = own + (1.0 - eps) * o0 + eps * o1;
      rv *= 0.5;
      remap[P + idx2] = rv;
    }
    double e = save[P + idx2];
    e += (1.0 - eps) * save[jremap + (size_t)g.jsize * k] + eps * save[jremapp1 + (size_t)g.jsize * k];
    e *= 0.5;
    emf[krow + fsj * j + nx + gw + (size_t)EMF_Y * N] = e;
  }
}

// ------------------------------------------------------------------------------------------------------------
// conservative update (gather form of the reference's scatter loop, same summation order per cell) + CT
// ------------------------------------------------------------------------------------------------------------
struct RotCoef { double lambda, ratio, alpha1, alpha2; };  // MHDRunGodunov.cpp:2039-2053

// What one cell's update reads, by value: the flat kernel loads it all from global memory, the z-marching one carries the
// plane k+1 entries over to the next plane (where they are the plane k entries).
struct UpdIn {
  double u[8];                                      // Uold of the cell
  double fx0[5], fy0[5], fz0[5];                    // fluxes through the cell's low faces
  double fx1[5], fy1[5], fz1[5];                    // ... of cells i+1, j+1, k+1 (= through the high faces)
  double eZ00, eZ10, eZ01, eZ11;                    // emf z at (i,j) (i+1,j) (i,j+1) (i+1,j+1), plane k
  double eY00, eY10, eY01, eY11;                    // emf y at (i,k) (i+1,k) (i,k+1) (i+1,k+1), row j
  double eX00, eX10, eX01, eX11;                    // emf x at (j,k) (j+1,k) (j,k+1) (j+1,k+1), column i
  double uA1, uB1, uC1;                             // Uold: IA of cell i+1, IB of cell j+1, IC of cell k+1 (CFL scan only)
};

// dt_slots != 0: the CFL scan of the NEW state rides along (compute_dt_mhd of the next step, MHDRunBase.cpp:140-250): an
// interior cell needs the new field on its three high faces, which belong to its +1 neighbours -- their CT update is
// repeated here from the same emf values (same expressions as below, hence the same bits) -- then the value of
// mhd_invdt_cell goes to one of RG_DT_SLOTS maxima.
template <bool ROT, bool GF>
RG_DEVFN void mhd_update3d_apply(const DevParams& g, const RotCoef rc, const IJK c, unsigned idx, const UpdIn& in,
                                 double* __restrict__ Unew, const double* __restrict__ remap, double dt, double dtdx, double dtdy,
                                 double dtdz, unsigned long long* dt_slots) {
  const size_t N = g.ncell;
  const int gw = g.gw;
  double u[8];
#pragma unroll
  for (int v = 0; v < 8; ++v) u[v] = in.u[v];
  const bool in_i = c.i >= gw && c.i < g.isize - gw, in_j = c.j >= gw && c.j < g.jsize - gw, in_k = c.k >= gw && c.k < g.ksize - gw;
  const bool shear = ROT && g.shearbox;
  if (in_i && in_j && in_k) {
    if (ROT) {  // Coriolis, before any flux is applied to this cell (MHDRunGodunov.cpp:2938-2945)
      const rg_recip_t inv_l = rg_recip(1.0 + rc.lambda);
      const double dsx = rg_div(2.0 * g.Omega0 * dt * u[IV], inv_l);
      const double dsy = rg_div(-0.5 * g.Omega0 * dt * u[IU], inv_l);
      u[IU] = u[IU] * rc.ratio + dsx;
      u[IV] = u[IV] * rc.ratio + dsy;
    }
    const double a1 = rc.alpha1, a2 = rc.alpha2;
    // contributions in the order the reference's k,j,i loop delivers them to this cell:
    // own iteration: +Fx, +Fy, +Fz ; then -Fx(i+1), -Fy(j+1), -Fz(k+1)
    const double* f = in.fx0;
    if (!(shear && c.i == gw)) u[ID] += f[ID] * dtdx;
    u[IP] += f[IP] * dtdx;
    if (ROT) { u[IU] += (a1 * f[IU] + a2 * f[IV]) * dtdx; u[IV] += (a1 * f[IV] - 0.25 * a2 * f[IU]) * dtdx; }
    else { u[IU] += f[IU] * dtdx; u[IV] += f[IV] * dtdx; }
    u[IW] += f[IW] * dtdx;
    f = in.fy0;   // y-normal frame: f[IU] is the y momentum flux, f[IV] the x momentum flux
    u[ID] += f[ID] * dtdy;
    u[IP] += f[IP] * dtdy;
    if (ROT) { u[IU] += (a1 * f[IV] + a2 * f[IU]) * dtdy; u[IV] += (a1 * f[IU] - 0.25 * a2 * f[IV]) * dtdy; }
    else { u[IU] += f[IV] * dtdy; u[IV] += f[IU] * dtdy; }
    u[IW] += f[IW] * dtdy;
    f = in.fz0;   // z-normal frame: f[IU] is the z momentum flux, f[IW] the x momentum flux
    u[ID] += f[ID] * dtdz;
    u[IP] += f[IP] * dtdz;
    if (ROT) { u[IU] += (a1 * f[IW] + a2 * f[IV]) * dtdz; u[IV] += (a1 * f[IV] - 0.25 * a2 * f[IW]) * dtdz; }
    else { u[IU] += f[IW] * dtdz; u[IV] += f[IV] * dtdz; }
    u[IW] += f[IU] * dtdz;
    f = in.fx1;
    if (!(shear && (c.i + 1) == (g.nx + gw))) u[ID] -= f[ID] * dtdx;
    u[IP] -= f[IP] * dtdx;
    if (ROT) { u[IU] -= (a1 * f[IU] + a2 * f[IV]) * dtdx; u[IV] -= (a1 * f[IV] - 0.25 * a2 * f[IU]) * dtdx; }
    else { u[IU] -= f[IU] * dtdx; u[IV] -= f[IV] * dtdx; }
    u[IW] -= f[IW] * dtdx;
    f = in.fy1;
    u[ID] -= f[ID] * dtdy;
    u[IP] -= f[IP] * dtdy;
    if (ROT) { u[IU] -= (a1 * f[IV] + a2 * f[IU]) * dtdy; u[IV] -= (a1 * f[IU] - 0.25 * a2 * f[IV]) * dtdy; }
    else { u[IU] -= f[IV] * dtdy; u[IV] -= f[IU] * dtdy; }
    u[IW] -= f[IW] * dtdy;
    f = in.fz1;
    u[ID] -= f[ID] * dtdz;
    u[IP] -= f[IP] * dtdz;
    if (ROT) { u[IU] -= (a1 * f[IW] + a2 * f[IV]) * dtdz; u[IV] -= (a1 * f[IV] - 0.25 * a2 * f[IW]) * dtdz; }
    else { u[IU] -= f[IW] * dtdz; u[IV] -= f[IV] * dtdz; }
    u[IW] -= f[IU] * dtdz;
    if (GF || g.grav_on) {  // momentum source before the shear remap of the density (MHDRunGodunov.cpp:3190-3192)
      const double rho_sum = in.u[ID] + u[ID];
      double gx, gy, gz;
      half_dt_gravity<GF>(g, idx, gx, gy, gz);
      u[IU] += gx * rho_sum;
      u[IV] += gy * rho_sum;
      u[IW] += gz * rho_sum;
    }
    if (shear) {  // remapped density flux at the two x borders, then the density floor (:3289-3300)
      const size_t P = (size_t)g.jsize * g.ksize;
      const size_t jk = (size_t)c.j + (size_t)g.jsize * c.k;
      if (c.i == gw) u[ID] += remap[jk];
      if (c.i == g.nx + gw - 1) u[ID] -= remap[P + jk];
      if (c.i == gw || c.i == g.nx + gw - 1) u[ID] = fmax(u[ID], g.smallr);
    }
  }
  // constrained transport on [gw, size-gw] in every direction (faces of the first high ghost layer included)
  const bool ct_i = c.i >= gw && c.i <= g.isize - gw, ct_j = c.j >= gw && c.j <= g.jsize - gw, ct_k = c.k >= gw && c.k <= g.ksize - gw;
  if (ct_i && ct_j && ct_k) {
    if (c.k < g.ksize - gw) {
      u[IA] += (in.eZ01 - in.eZ00) * dtdy;
      u[IB] -= (in.eZ10 - in.eZ00) * dtdx;
    }
    u[IA] -= (in.eY01 - in.eY00) * dtdz;
    u[IB] += (in.eX01 - in.eX00) * dtdz;
    u[IC] += (in.eY10 - in.eY00) * dtdx;
    u[IC] -= (in.eX10 - in.eX00) * dtdy;
  }
#pragma unroll
  for (int v = 0; v < 8; ++v) RG_STREAM_STORE(&Unew[idx + v * N], u[v]);
  if (dt_slots && in_i && in_j && in_k) {
    double bnx = in.uA1;     // low x face of cell i+1
    bnx += (in.eZ11 - in.eZ10) * dtdy;          // (k < ksize - gw holds for an interior cell)
    bnx -= (in.eY11 - in.eY10) * dtdz;
    double bny = in.uB1;     // low y face of cell j+1
    bny -= (in.eZ11 - in.eZ01) * dtdx;
    bny += (in.eX11 - in.eX10) * dtdz;
    double bnz = in.uC1;     // low z face of cell k+1
    bnz += (in.eY11 - in.eY01) * dtdx;
    bnz -= (in.eX11 - in.eX01) * dtdy;
    const Prim8 q = mhd_prim(g, u, bnx, bny, bnz, 0.0);
    double sx, sy, sz;
    info_speeds(g, q, sx, sy, sz);
    if (g.Omega0 > 0) sy += 1.5 * g.Omega0 * g.deltaX / 2;  // shear velocity at the box edge (MHDRunBase.cpp:223-225)
    rgpu::rg_slot_max(dt_slots + ((idx >> 6) & (rgpu::RG_DT_SLOTS - 1)), sx / g.dx + sy / g.dy + sz / g.dz);
  }
}

// One thread per column (i, j) and z segment: planes [k_lo + seg * seg_len, + seg_len) of [k_lo, k_hi), marching upwards.  The
// entries of plane k+1 a cell reads -- Fz, emf x and y at two positions each, the z face field for the CFL scan: 10 of its 53
// loads, and the ones no cache keeps when a one-thread-per-cell kernel walks plane after plane -- are loaded once and carried in
// registers to the next plane, where they are the cell's own.  Short segments: on gfx950 a wave's loads of plane k+1 return
// behind its stores of plane k (one in-order counter), so long marches expose the write latency -- 512^3, same box: one cell per
// thread 8.07 ms, segments of 2 / 3 / 4 / 8 / 32 planes 7.49 / 7.42 / 7.50 / 7.65 / 9.0.  t = seg * (isize * jsize) + column.
template <bool ROT, bool GF>
RG_DEVFN void mhd_update3d_column(const DevParams& g, const RotCoef rc, const double* __restrict__ Uold,
                                  double* __restrict__ Unew, const double* __restrict__ F, const double* __restrict__ emf,
                                  const double* __restrict__ remap, double dt, double dtdx, double dtdy, double dtdz,
                                  unsigned t, int k_lo, int k_hi, int seg_len, unsigned long long* dt_slots = 0) {
  const size_t N = g.ncell, NF = g.fN;
  const unsigned sj = g.sj, sk = g.sk, fsj = g.fsj, fsk = g.fsk;   // (F / emf have a pitch of their own: flux_index)
  const int gw = g.gw;
  const unsigned seg = t / sk, col = t - seg * sk;
  IJK c = unflatten(g, col);
  const int ka = k_lo + (int)seg * seg_len;
  const int kb = (ka + seg_len < k_hi) ? ka + seg_len : k_hi;
  unsigned idx = col + (unsigned)ka * sk;
  unsigned fidx = flux_index(g, c.i, c.j, ka);
  const bool in_col = c.i >= gw && c.i < g.isize - gw && c.j >= gw && c.j < g.jsize - gw;
  const bool ct_col = c.i >= gw && c.i <= g.isize - gw && c.j >= gw && c.j <= g.jsize - gw;
  if (!ct_col) {   // ghost columns: the new array gets the old values (refilled by the next ghost fill)
    for (int k = ka; k < kb; ++k, idx += sk) {
#pragma unroll
      for (int v = 0; v < 8; ++v) RG_STREAM_STORE(&Unew[idx + v * N], Uold[idx + v * N]);
    }
    return;
  }
  const double* eZ = emf + (size_t)EMF_Z * NF; const double* eY = emf + (size_t)EMF_Y * NF; const double* eX = emf + (size_t)EMF_X * NF;
  // this plane's share of the carried entries
  double fz[5], eX00, eX10, eY00, eY10, uC;
#pragma unroll
  for (int v = 0; v < 5; ++v) fz[v] = in_col ? F[fidx + (size_t)(F_Z + v) * NF] : 0.0;
  eX00 = eX[fidx]; eX10 = eX[fidx + fsj]; eY00 = eY[fidx]; eY10 = eY[fidx + 1];
  uC = Uold[idx + IC * N];
  for (int k = ka; k < kb; ++k, idx += sk, fidx += fsk) {
    c.k = k;
    UpdIn in;
    const bool up = k + 1 < g.ksize;   // (the top plane is a ghost plane: nothing of plane k+1 is used there)
    const unsigned nx_ = idx + sk, fnx = fidx + fsk;
#pragma unroll
    for (int v = 0; v < 8; ++v) in.u[v] = (v == IC) ? uC : Uold[idx + v * N];
#pragma unroll
    for (int v = 0; v < 5; ++v) {
      in.fx0[v] = in_col ? F[fidx + (size_t)(F_X + v) * NF] : 0.0;
      in.fy0[v] = in_col ? F[fidx + (size_t)(F_Y + v) * NF] : 0.0;
      in.fz0[v] = fz[v];
      in.fx1[v] = in_col ? F[fidx + 1 + (size_t)(F_X + v) * NF] : 0.0;
      in.fy1[v] = in_col ? F[fidx + fsj + (size_t)(F_Y + v) * NF] : 0.0;
      in.fz1[v] = (in_col && up) ? F[fnx + (size_t)(F_Z + v) * NF] : 0.0;
    }
    in.eZ00 = eZ[fidx]; in.eZ10 = eZ[fidx + 1]; in.eZ01 = eZ[fidx + fsj]; in.eZ11 = eZ[fidx + fsj + 1];
    in.eY00 = eY00; in.eY10 = eY10;
    in.eX00 = eX00; in.eX10 = eX10;
    in.eY01 = up ? eY[fnx] : 0.0; in.eY11 = up ? eY[fnx + 1] : 0.0;
    in.eX01 = up ? eX[fnx] : 0.0; in.eX11 = up ? eX[fnx + fsj] : 0.0;
    in.uA1 = Uold[idx + 1 + IA * N];
    in.uB1 = Uold[idx + sj + IB * N];
    in.uC1 = up ? Uold[nx_ + IC * N] : 0.0;
    mhd_update3d_apply<ROT, GF>(g, rc, c, idx, in, Unew, remap, dt, dtdx, dtdy, dtdz, dt_slots);
#pragma unroll
    for (int v = 0; v < 5; ++v) fz[v] = in.fz1[v];
    eX00 = in.eX01; eX10 = in.eX11; eY00 = in.eY01; eY10 = in.eY11; uC = in.uC1;
  }
}

}  // namespace rgpu_dev
