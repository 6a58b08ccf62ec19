// kernels_dissipative.h -- the operator-split dissipative stage after the Godunov update ([hydro] nu, [MHD] eta):
//   visc_flux_cell      U -> Fd        viscous stress fluxes at the low faces     HydroRunBase.cpp:431-560, 876-1160
//   flux_update_cell    U += dFd       compute_hydro_update(_energy)              HydroRunBase.cpp:1461-1533, 1633-1700
//   resist_emf_cell     U -> E         -eta curl B at the low edges               MHDRunBase.cpp:455-575
//   resist_ct_cell      U(B) += curl E compute_ct_update_{2d,3d}                  MHDRunBase.cpp:256-344
//   resist_eflux_cell   U -> Fd[IP]    -eta (J x B).n at the low faces            MHDRunBase.cpp:697-900
// Fd holds NVH = ND + 2 components (rho, E, momenta) per direction: slot (D * NVH + v).  All kernels run on the state
// the step has just written; its ghosts are refilled first (call sites: HydroRunGodunov.cpp:2620-2640,
// mhd_godunov_unsplit_cpu_v3.cpp:662-694, MHDRunGodunov.cpp:3379-3420).  Direction-generic, one thread per cell.
#pragma once
#include "kernels_mhd3d.h"

namespace rgpu_dev {

RG_DEVFN double cell_vel(const double* __restrict__ U, size_t N, unsigned o, int a) { return U[o + (size_t)(IU + a) * N] / U[o + (size_t)ID * N]; }

// Range of the resistive emf and of its CT update.  At a slab interface the ghost planes are the neighbour's interior:
// its plane next to the interface gets the same CT update there, and the energy flux of the cells on this side reads
// that field, so the range is extended by one plane into the ghosts (the emf by one more) -- the slab run then equals
// the single-domain run bit for bit.  At physical faces the reference's ranges are kept.
RG_DEVFN bool in_resist_range(const DevParams& g, const IJK c, int ND, int extra) {
  if (c.i < g.gw || c.i > g.isize - g.gw || c.j < g.gw || c.j > g.jsize - g.gw) return false;
  if (ND == 3) {
    const int lo = g.gw - (g.zlo_copy ? 1 : 0), hi = g.ksize - g.gw + (g.zhi_copy ? extra : 0);
    if (c.k < lo || c.k > hi) return false;
  }
  return true;
}

RG_DEVFN bool in_face_range(const DevParams& g, const IJK c, int ND) {
  if (c.i < g.gw || c.i > g.isize - g.gw || c.j < g.gw || c.j > g.jsize - g.gw) return false;
  if (ND == 3 && (c.k < g.gw || c.k > g.ksize - g.gw)) return false;
  return true;
}

template <int ND>
RG_DEVFN void visc_flux_cell(const DevParams& g, const double* __restrict__ U, double* __restrict__ Fd, double nu, double dt,
                             unsigned idx) {
  const IJK c = unflatten(g, idx);
  if (!in_face_range(g, c, ND)) return;
  const int NVH = ND + 2;
  const size_t N = g.ncell;
  const unsigned st[3] = {1u, g.sj, g.sk};
  const double h[3] = {g.dx, g.dy, g.dz};
  const double two3rd = 2. / 3.;
#pragma unroll
  for (int D = 0; D < ND; ++D) {
    const unsigned o = idx, oL = idx - st[D];
    const double rho = 0.5 * (U[o + ID * N] + U[oL + ID * N]);
    double vo[ND], vl[ND], uavg[ND], grad[ND][ND];   // grad[b][a] = d u_a / d x_b at the face
#pragma unroll
    for (int a = 0; a < ND; ++a) { vo[a] = cell_vel(U, N, o, a); vl[a] = cell_vel(U, N, oL, a); }
#pragma unroll
    for (int a = 0; a < ND; ++a) { uavg[a] = 0.5 * (vo[a] + vl[a]); grad[D][a] = (vo[a] - vl[a]) / h[D]; }
#pragma unroll
    for (int T = 0; T < ND; ++T) {
      if (T == D) continue;
#pragma unroll
      for (int a = 0; a < ND; ++a) {
        if (a != D && a != T) continue;
        const double uR = cell_vel(U, N, o + st[T], a) + cell_vel(U, N, oL + st[T], a);
        const double uL = cell_vel(U, N, o - st[T], a) + cell_vel(U, N, oL - st[T], a);
        grad[T][a] = (uR - uL) / h[T] / 4;
      }
    }
    double t[ND];
    {
      double tr = 2.0 * grad[D][D];
#pragma unroll
      for (int T = 0; T < ND; ++T) if (T != D) tr = tr - grad[T][T];
      t[D] = -two3rd * nu * rho * tr;
    }
#pragma unroll
    for (int T = 0; T < ND; ++T) {
      if (T == D) continue;
      const int a = (D < T) ? D : T, b = (D < T) ? T : D;
      t[T] = -nu * rho * (grad[b][a] + grad[a][b]);
    }
    double* f = Fd + (size_t)(D * NVH) * N + idx;
    f[ID * N] = 0.0;
#pragma unroll
    for (int a = 0; a < ND; ++a) f[(size_t)(IU + a) * N] = t[a] * dt / h[D];
    if (g.cIso <= 0) {
      double e = uavg[0] * t[0];
#pragma unroll
      for (int a = 1; a < ND; ++a) e = e + uavg[a] * t[a];
      f[IP * N] = e * dt / h[D];
    } else {
      f[IP * N] = 0.0;
    }
  }
}

// U(v) += (Fx(i) - Fx(i+1)); += (Fy(j) - Fy(j+1)); [+= (Fz(k) - Fz(k+1))] for v in [v0, v1), interior cells
template <int ND>
RG_DEVFN void flux_update_cell(const DevParams& g, double* __restrict__ U, const double* __restrict__ Fd, int v0, int v1, unsigned idx) {
  const IJK c = unflatten(g, idx);
  if (c.i < g.gw || c.i >= g.isize - g.gw || c.j < g.gw || c.j >= g.jsize - g.gw) return;
  if (ND == 3 && (c.k < g.gw || c.k >= g.ksize - g.gw)) return;
  const int NVH = ND + 2;
  const size_t N = g.ncell;
  const unsigned st[3] = {1u, g.sj, g.sk};
  for (int v = v0; v < v1; ++v) {
    double u = U[idx + (size_t)v * N];
#pragma unroll
    for (int D = 0; D < ND; ++D) {
      const double* f = Fd + (size_t)(D * NVH + v) * N;
      u += (f[idx] - f[idx + st[D]]);
    }
    U[idx + (size_t)v * N] = u;
  }
}

// E slots follow EmfIndex (EMF_Z = 0, EMF_Y = 1, EMF_X = 2); 2D has EMF_Z only
template <int ND>
RG_DEVFN void resist_emf_cell(const DevParams& g, const double* __restrict__ U, double* __restrict__ E, double eta, unsigned idx) {
  const IJK c = unflatten(g, idx);
  if (!in_resist_range(g, c, ND, 1)) return;
  const size_t N = g.ncell;
  const unsigned sj = g.sj, sk = g.sk;
  const double* A = U + IA * N; const double* B = U + IB * N; const double* C = U + IC * N;
  const double dbydx = (B[idx] - B[idx - 1]) / g.dx;
  const double dbxdy = (A[idx] - A[idx - sj]) / g.dy;
  if (ND == 3) {
    const double dbzdx = (C[idx] - C[idx - 1]) / g.dx;
    const double dbzdy = (C[idx] - C[idx - sj]) / g.dy;
    const double dbxdz = (A[idx] - A[idx - sk]) / g.dz;
    const double dbydz = (B[idx] - B[idx - sk]) / g.dz;
    E[idx + (size_t)EMF_X * N] = -eta * (dbzdy - dbydz);
    E[idx + (size_t)EMF_Y * N] = -eta * (dbxdz - dbzdx);
  }
  E[idx + (size_t)EMF_Z * N] = -eta * (dbydx - dbxdy);
}

template <int ND>
RG_DEVFN void resist_ct_cell(const DevParams& g, double* __restrict__ U, const double* __restrict__ E, double dtdx, double dtdy,
                             double dtdz, unsigned idx) {
  const IJK c = unflatten(g, idx);
  if (!in_resist_range(g, c, ND, 0)) return;
  const size_t N = g.ncell;
  const unsigned sj = g.sj, sk = g.sk;
  const double* eZ = E + (size_t)EMF_Z * N; const double* eY = E + (size_t)EMF_Y * N; const double* eX = E + (size_t)EMF_X * N;
  double a = U[idx + IA * N], b = U[idx + IB * N];
  if (ND == 2) {
    a += (eZ[idx + sj] - eZ[idx]) * dtdy;
    b -= (eZ[idx + 1] - eZ[idx]) * dtdx;
  } else {
    double cc = U[idx + IC * N];
    if (c.k < g.ksize - g.gw || g.zhi_copy) {
      a += (eZ[idx + sj] - eZ[idx]) * dtdy;
      b -= (eZ[idx + 1] - eZ[idx]) * dtdx;
    }
    a -= (eY[idx + sk] - eY[idx]) * dtdz;
    b += (eX[idx + sk] - eX[idx]) * dtdz;
    cc += (eY[idx + 1] - eY[idx]) * dtdx;
    cc -= (eX[idx + sj] - eX[idx]) * dtdy;
    U[idx + IC * N] = cc;
  }
  U[idx + IA * N] = a;
  U[idx + IB * N] = b;
}

template <int ND>
RG_DEVFN void resist_eflux_cell(const DevParams& g, const double* __restrict__ U, double* __restrict__ Fd, double eta, double dt,
                                unsigned idx) {
  const IJK c = unflatten(g, idx);
  if (!in_face_range(g, c, ND)) return;
  const int NVH = ND + 2;
  const size_t N = g.ncell;
  const unsigned o = idx, sj = g.sj, sk = g.sk;
  const double dx = g.dx, dy = g.dy, dz = g.dz;
  const double* A = U + IA * N; const double* B = U + IB * N; const double* C = U + IC * N;
  double bx, by, bz, jx, jy, jz, jxp1, jyp1, jzp1;
  double* fx = Fd + (size_t)(0 * NVH + IP) * N; double* fy = Fd + (size_t)(1 * NVH + IP) * N;
  if (ND == 2) {
    by = (B[o] + B[o - 1] + B[o + sj] + B[o - 1 + sj]) / 4;
    bz = (C[o] + C[o - 1]) / 2;
    jy = -(C[o] - C[o - 1]) / dx;
    jz = (B[o] - B[o - 1]) / dx - (A[o] - A[o - sj]) / dy;
    jzp1 = (B[o + sj] - B[o - 1 + sj]) / dx - (A[o + sj] - A[o]) / dy;
    jz = (jz + jzp1) / 2;
    fx[o] = -eta * (jy * bz - jz * by) * dt / dx;
    bx = (A[o] + A[o - sj] + A[o + 1] + A[o + 1 - sj]) / 4;
    bz = (C[o] + C[o - sj]) / 2;
    jx = (C[o] - C[o - sj]) / dy;
    jz = (B[o] - B[o - 1]) / dx - (A[o] - A[o - sj]) / dy;
    jzp1 = (B[o + 1] - B[o]) / dx - (A[o + 1] - A[o + 1 - sj]) / dy;
    jz = (jz + jzp1) / 2;
    fy[o] = -eta * (jz * bx - jx * bz) * dt / dy;
  } else {
    double* fz = Fd + (size_t)(2 * NVH + IP) * N;
    by = (B[o] + B[o - 1] + B[o + sj] + B[o - 1 + sj]) / 4;
    bz = (C[o] + C[o - 1] + C[o + sk] + C[o - 1 + sk]) / 4;
    jy = (A[o] - A[o - sk]) / dz - (C[o] - C[o - 1]) / dx;
    jyp1 = (A[o + sk] - A[o]) / dz - (C[o + sk] - C[o - 1 + sk]) / dx;
    jy = (jy + jyp1) / 2;
    jz = (B[o] - B[o - 1]) / dx - (A[o] - A[o - sj]) / dy;
    jzp1 = (B[o + sj] - B[o - 1 + sj]) / dx - (A[o + sj] - A[o]) / dy;
    jz = (jz + jzp1) / 2;
    fx[o] = -eta * (jy * bz - jz * by) * dt / dx;
    bx = (A[o] + A[o - sj] + A[o + 1] + A[o + 1 - sj]) / 4;
    bz = (C[o] + C[o - sj] + C[o + sk] + C[o - sj + sk]) / 4;
    jx = (C[o] - C[o - sj]) / dy - (B[o] - B[o - sk]) / dz;
    jxp1 = (C[o + sk] - C[o - sj + sk]) / dy - (B[o + sk] - B[o]) / dz;
    jx = (jx + jxp1) / 2;
    jz = (B[o] - B[o - 1]) / dx - (A[o] - A[o - sj]) / dy;
    jzp1 = (B[o + 1] - B[o]) / dx - (A[o + 1] - A[o + 1 - sj]) / dy;
    jz = (jz + jzp1) / 2;
    fy[o] = -eta * (jz * bx - jx * bz) * dt / dy;
    bx = (A[o] + A[o - sk] + A[o + 1] + A[o + 1 - sk]) / 4;
    by = (B[o] + B[o - sk] + B[o + sj] + B[o + sj - sk]) / 4;
    jx = (C[o] - C[o - sj]) / dy - (B[o] - B[o - sk]) / dz;
    jxp1 = (C[o + sj] - C[o]) / dy - (B[o + sj] - B[o + sj - sk]) / dz;
    jx = (jx + jxp1) / 2;
    jy = (A[o] - A[o - sk]) / dz - (C[o] - C[o - 1]) / dx;
    jyp1 = (A[o + 1] - A[o + 1 - sk]) / dz - (C[o + 1] - C[o]) / dx;
    jy = (jy + jyp1) / 2;
    fz[o] = -eta * (jx * by - jy * bx) * dt / dz;
  }
}

}  // namespace rgpu_dev
