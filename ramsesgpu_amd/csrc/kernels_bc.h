// kernels_bc.h -- ghost-cell fill, jet inflow, shearing-box ghost remap and the CFL scan, per-thread bodies.
//   bc_face_cell        make_boundary2<bct,loc>          make_boundary_base.h:1040-1332
//   jet_cell            make_jet                         HydroRunBase.cpp:2374-2408
//   shear_ghost_cell    make_boundaries_shear            MHDRunGodunov.cpp:3539-3759
//   hydro/mhd_invdt     compute_dt / compute_dt_mhd      HydroRunBase.cpp:372-426 / MHDRunBase.cpp:140-250
#pragma once
#include "ou_forcing.h"
#include "kernels_mhd3d.h"

namespace rgpu_dev {

// One thread per ghost cell of one face; all variables in a loop.  The thread index is decoded so that x stays
// the fastest-varying coordinate whenever the face is not an x face (coalesced rows).
// dir 0,1,2 ; side 0 = min, 1 = max ; bct = 1 dirichlet, 2 neumann, 3 periodic
RG_DEVFN void bc_face_cell(const DevParams& g, double* __restrict__ U, int dir, int side, int bct, unsigned idx) {
  const int gw = g.gw;
  int a, i, j, k;  // a = ghost layer 0..gw-1
  if (dir == 0) {
    a = (int)(idx % (unsigned)gw);
    const unsigned t = idx / (unsigned)gw;
    j = (int)(t % (unsigned)g.jsize);
    k = (int)(t / (unsigned)g.jsize);
    i = 0;
  } else if (dir == 1) {
    i = (int)(idx % (unsigned)g.isize);
    const unsigned t = idx / (unsigned)g.isize;
    a = (int)(t % (unsigned)gw);
    k = (int)(t / (unsigned)gw);
    j = 0;
  } else {
    i = (int)(idx % (unsigned)g.isize);
    const unsigned t = idx / (unsigned)g.isize;
    j = (int)(t % (unsigned)g.jsize);
    a = (int)(t / (unsigned)g.jsize);
    k = 0;
  }
  const int n = (dir == 0) ? g.nx : (dir == 1) ? g.ny : g.nz;
  const int ghost = (side == 0) ? a : n + gw + a;
  int src;
  if (bct == 1) src = (side == 0) ? 2 * gw - 1 - ghost : 2 * n + 2 * gw - 1 - ghost;
  else if (bct == 2) src = (side == 0) ? gw : n + gw - 1;
  else src = (side == 0) ? n + ghost : ghost - n;
  int io = i, jo = j, ko = k, ii = i, ji = j, ki = k;
  if (dir == 0) { io = ghost; ii = src; } else if (dir == 1) { jo = ghost; ji = src; } else { ko = ghost; ki = src; }
  const size_t N = g.ncell;
  const size_t o_out = (size_t)io + (size_t)g.sj * jo + (size_t)g.sk * ko;
  const size_t o_in = (size_t)ii + (size_t)g.sj * ji + (size_t)g.sk * ki;
  const int normal_mom = (dir == 0) ? IU : (dir == 1) ? IV : IW;
  for (int v = 0; v < g.nvar; ++v) {
    const double sign = (bct == 1 && v == normal_mom) ? -1.0 : 1.0;
    U[o_out + v * N] = U[o_in + v * N] * sign;
  }
}

// BC_Z_STRATIFIED (make_boundary2_z_stratified_cpu, make_boundary_base.h:1356-1647), ghost width 3: one thread per
// (i,j) column of one z face.  Density extrapolated along the isothermal hydrostatic profile (ratios r1..r3, computed on
// the host with the C library's exp()), tangential velocities kept, normal momentum outflow-only, tangential field zero,
// the energy left alone.  The reference's second loop derives the ghost Bz from div B = 0 with the tangential field it
// has just set to zero on every column: the three corrections are (0 - 0)/dx terms, written out here as such.
struct ZStrat { double r1, r2, r3; };

RG_DEVFN void zstrat_column(const DevParams& g, const ZStrat zs, double* __restrict__ U, int side, unsigned ij) {
  const size_t N = g.ncell;
  const size_t sk = g.sk;
  const int i = (int)(ij % (unsigned)g.isize), j = (int)(ij / (unsigned)g.isize);
  // planes: src = the last interior plane, gh[0..2] = ghost planes going outwards
  const int src = (side == 0) ? 3 : g.ksize - 4;
  const int step = (side == 0) ? -1 : +1;
  const size_t o_src = (size_t)ij + sk * (size_t)src;
  size_t o_gh[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) o_gh[a] = (size_t)ij + sk * (size_t)(src + step * (a + 1));
  const double rho_s = U[o_src + ID * N];
  double rho[3];
  rho[0] = rho_s * zs.r1;
  rho[1] = rho_s * zs.r1 * zs.r2;
  rho[2] = rho_s * zs.r1 * zs.r2 * zs.r3;
#pragma unroll
  for (int a = 0; a < 3; ++a) U[o_gh[a] + ID * N] = rho[a];
  const double mu = U[o_src + IU * N], mv = U[o_src + IV * N], mw = U[o_src + IW * N];
  const double w = (side == 0) ? fmin(mw, 0.0) : fmax(mw, 0.0);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    U[o_gh[a] + IU * N] = mu / rho_s * rho[a];
    U[o_gh[a] + IV * N] = mv / rho_s * rho[a];
    U[o_gh[a] + IW * N] = w;
    U[o_gh[a] + IA * N] = 0.0;
    U[o_gh[a] + IB * N] = 0.0;
  }
  if (i < g.isize - 1 && j < g.jsize - 1) {
    const double zero = 0.0;
    const double dbz = g.dz * ((zero - zero) / g.dx + (zero - zero) / g.dy);
    if (side == 0) {   // Bz lives on the low face: ghost planes 2,1,0 from the interior face 3
      const double bz = U[o_src + IC * N];
      U[o_gh[0] + IC * N] = bz + dbz;
      U[o_gh[1] + IC * N] = bz + dbz + dbz;
      U[o_gh[2] + IC * N] = bz + dbz + dbz + dbz;
    } else {           // the face ksize-3 is evolved by the CT update; two faces beyond it
      const double bz = U[o_gh[0] + IC * N];
      U[o_gh[1] + IC * N] = bz - dbz;
      U[o_gh[2] + IC * N] = bz - dbz - dbz;
    }
  }
}

struct JetParams { int ijet, offsetJet; double djet, ejet, mjet; };  // ejet = pjet/(gamma0-1)+0.5*djet*ujet^2, mjet = djet*ujet

// one thread per injected ghost cell: 2D idx over ijet*gw (low-y rows), 3D over ijet*ijet*gw (low-z planes)
RG_DEVFN void jet_cell(const DevParams& g, const JetParams jp, double* __restrict__ U, unsigned idx) {
  const size_t N = g.ncell;
  const int gw = g.gw;
  if (!g.three_d) {
    const int ii = (int)(idx % (unsigned)jp.ijet), j = (int)(idx / (unsigned)jp.ijet);
    const size_t o = (size_t)(gw + jp.offsetJet + ii) + (size_t)g.sj * j;
    U[o + ID * N] = jp.djet; U[o + IP * N] = jp.ejet; U[o + IU * N] = 0.0; U[o + IV * N] = jp.mjet;
  } else {
    const int ii = (int)(idx % (unsigned)jp.ijet);
    const unsigned t = idx / (unsigned)jp.ijet;
    const int jj = (int)(t % (unsigned)jp.ijet), k = (int)(t / (unsigned)jp.ijet);
    const size_t o = (size_t)(gw + jp.offsetJet + ii) + (size_t)g.sj * (gw + jp.offsetJet + jj) + (size_t)g.sk * k;
    U[o + ID * N] = jp.djet; U[o + IP * N] = jp.ejet; U[o + IU * N] = 0.0; U[o + IV * N] = 0.0; U[o + IW * N] = jp.mjet;
  }
}

// limited y slope of the copied border columns (MHDRunGodunov.cpp:3609-3619): note dcen = (dlft+drgt)/2/slope_type
RG_DEVFN double shear_border_slope(double st, double bm, double b0, double bp) {
  if (!(st == 1 || st == 2)) return 0.0;
  const double dlft = st * (b0 - bm);
  const double drgt = st * (bp - b0);
  const double dcen = 0.5 * (dlft + drgt) / st;
  const double dsgn = (dcen >= 0.0) ? 1.0 : -1.0;
  const double slop = fmin(fabs(dlft), fabs(drgt));
  const double dlim = ((dlft * drgt) <= 0.0) ? 0.0 : slop;
  return dsgn * fmin(dlim, fabs(dcen));
}

struct ShearGhost { int jplus; double eps_min, eps_max; };  // from totalTime+dt on the host (:3554-3557)

// One thread per (ghost column i in 0..gw-1, interior row j, any plane k); fills BOTH x sides.  The reference's
// border copies and slope arrays are never materialised: sources are the gw innermost interior columns, read
// directly (they are disjoint from the ghost columns being written as long as nx >= gw).
RG_DEVFN void shear_ghost_cell(const DevParams& g, const ShearGhost sg, double* __restrict__ U, unsigned idx) {
  const int gw = g.gw, nx = g.nx, ny = g.ny;
  const int i = (int)(idx % (unsigned)gw);
  const unsigned t = idx / (unsigned)gw;
  const int j = gw + (int)(t % (unsigned)ny);
  const int k = (int)(t / (unsigned)ny);
  const size_t N = g.ncell;
  const size_t krow = (size_t)g.sk * k;
  const double st = g.slope_type;
  // ---- inner (xmin) ghosts <- outer interior columns (nx + i), shifted by -(jplus+1) ----
  {
    int jr = j - sg.jplus - 1, jr1 = jr + 1;
    const double eps = sg.eps_min;
    if (jr < gw) jr += ny;
    if (jr1 < gw) jr1 += ny;
    const double lambda = 0.5 * eps * (eps - 1.0);
    const size_t col = krow + (size_t)(nx + i);
    for (int v = 0; v < 8; ++v) {
      const double* b = U + col + v * N;
      double out;
      if (v == IB) {
        // one-sided difference; like every slope it exists only for slope_type 1 or 2 (:3585-3604)
        const double slope = (st == 1 || st == 2) ? b[(size_t)g.sj * (jr + 1)] - b[(size_t)g.sj * jr] : 0.0;
        out = b[(size_t)g.sj * jr] + eps * slope;
      } else {
        const double s0 = shear_border_slope(st, b[(size_t)g.sj * (jr - 1)], b[(size_t)g.sj * jr], b[(size_t)g.sj * (jr + 1)]);
        const double s1 = shear_border_slope(st, b[(size_t)g.sj * (jr1 - 1)], b[(size_t)g.sj * jr1], b[(size_t)g.sj * (jr1 + 1)]);
        out = (1.0 - eps) * b[(size_t)g.sj * jr] + eps * b[(size_t)g.sj * jr1] + lambda * (s0 - s1);
      }
      U[krow + (size_t)g.sj * j + i + v * N] = out;
    }
  }
  // ---- outer (xmax) ghosts <- inner interior columns (gw + i), shifted by +jplus ----
  {
    int jr = j + sg.jplus, jr1 = jr + 1;
    const double eps = sg.eps_max;
    if (jr > ny + gw - 1) jr -= ny;
    if (jr1 > ny + gw - 1) jr1 -= ny;
    const double lambda = 0.5 * eps * (eps - 1.0);
    const size_t col = krow + (size_t)(gw + i);
    for (int v = 0; v < 8; ++v) {
      if (v == IA && i == 0) continue;  // the first outer Bx ghost is an evolved face value: keep it (:3727-3735)
      const double* b = U + col + v * N;
      double out;
      if (v == IB) {
        // one-sided difference; like every slope it exists only for slope_type 1 or 2 (:3585-3604)
        const double slope = (st == 1 || st == 2) ? b[(size_t)g.sj * (jr + 1)] - b[(size_t)g.sj * jr] : 0.0;
        out = b[(size_t)g.sj * jr] + eps * slope;
      } else {
        const double s0 = shear_border_slope(st, b[(size_t)g.sj * (jr - 1)], b[(size_t)g.sj * jr], b[(size_t)g.sj * (jr + 1)]);
        const double s1 = shear_border_slope(st, b[(size_t)g.sj * (jr1 - 1)], b[(size_t)g.sj * jr1], b[(size_t)g.sj * (jr1 + 1)]);
        out = (1.0 - eps) * b[(size_t)g.sj * jr] + eps * b[(size_t)g.sj * jr1] + lambda * (s1 - s0);
      }
      U[krow + (size_t)g.sj * j + (nx + gw + i) + v * N] = out;
    }
  }
}

// ---- the in-plane ghost fill of whole z planes in ONE pass -------------------------------------------------------
// make_all_boundaries fills x faces, then y faces over the full x extent, so that corner cells are images of images
// (make_boundary_base.h:1040-1332; shearing box: Y, shear remap of the x borders, Y -- MHDRunGodunov.cpp:3779-3793).  Every value
// these passes leave in a ghost cell of plane k is a function of INTERIOR cells of plane k alone:
//   x ghost, interior row      X fill of (i, j)                                    | shearing box: the remap formula at (i, j)
//   y ghost, interior column   Y fill: the image row src_j                         | periodic y: the wrapped row
//   corner                     Y fill of the X-filled cell = U(src_i, src_j) s_x s_y | the remap formula at (i, wrapped row)
// (s = -1 on the normal momentum of a reflecting face: exact), so one thread per ghost cell writes it from interior reads only:
// one launch instead of two (plain) or three (shearing box), no ordering between threads.  The shearing-box form needs periodic
// y faces: the remap reads the rows jr - 1 .. jr1 + 1 around its shifted rows, which the first Y pass had wrapped periodically.
struct FillXY { int bx0, bx1, by0, by1, shear; ShearGhost sg; };   // face types at xmin, xmax, ymin, ymax (plain: 1, 2, 3)

RG_DEVFN int fill_src_index(int bct, int side, int n, int gw, int ghost) {   // source index of bc_face_cell
  if (bct == 1) return (side == 0) ? 2 * gw - 1 - ghost : 2 * n + 2 * gw - 1 - ghost;
  if (bct == 2) return (side == 0) ? gw : n + gw - 1;
  return (side == 0) ? n + ghost : ghost - n;
}
RG_DEVFN int wrap_row(int r, int gw, int ny) { return r < gw ? r + ny : (r >= ny + gw ? r - ny : r); }

// value of variable v in the shearing-box x ghost column (side 0: column i, side 1: column nx + gw + i; i in 0..gw-1) at INTERIOR row j
// of the plane starting at krow: the expressions of shear_ghost_cell, its border rows read with the periodic wrap in y
RG_DEVFN double shear_ghost_value(const DevParams& g, const ShearGhost sg, const double* __restrict__ U, int side, int i, int j, size_t krow, int v) {
  const int gw = g.gw, nx = g.nx, ny = g.ny;
  const size_t N = g.ncell;
  const double st = g.slope_type;
  int jr, jr1;
  double eps;
  size_t col;
  if (side == 0) {   // inner (xmin) ghosts <- outer interior columns (nx + i), shifted by -(jplus+1)
    jr = j - sg.jplus - 1; jr1 = jr + 1; eps = sg.eps_min;
    if (jr < gw) jr += ny;
    if (jr1 < gw) jr1 += ny;
    col = krow + (size_t)(nx + i);
  } else {           // outer (xmax) ghosts <- inner interior columns (gw + i), shifted by +jplus
    jr = j + sg.jplus; jr1 = jr + 1; eps = sg.eps_max;
    if (jr > ny + gw - 1) jr -= ny;
    if (jr1 > ny + gw - 1) jr1 -= ny;
    col = krow + (size_t)(gw + i);
  }
  const double lambda = 0.5 * eps * (eps - 1.0);
  const double* b = U + col + v * N;
#define RG_ROW(r) b[(size_t)g.sj * wrap_row((r), gw, ny)]
  if (v == IB) {
    const double slope = (st == 1 || st == 2) ? RG_ROW(jr + 1) - RG_ROW(jr) : 0.0;
    return RG_ROW(jr) + eps * slope;
  }
  const double s0 = shear_border_slope(st, RG_ROW(jr - 1), RG_ROW(jr), RG_ROW(jr + 1));
  const double s1 = shear_border_slope(st, RG_ROW(jr1 - 1), RG_ROW(jr1), RG_ROW(jr1 + 1));
  const double d = (side == 0) ? (s0 - s1) : (s1 - s0);
  return (1.0 - eps) * RG_ROW(jr) + eps * RG_ROW(jr1) + lambda * d;
#undef RG_ROW
}

// ghost cells of one plane: 2 gw full rows (isize cells each), then 2 gw cells of each of the ny interior rows
// (host side: 2 gw (isize + ny) threads per plane)

// t = ghost cell of the plane (see above), k = its plane
RG_DEVFN void fill_xy_cell(const DevParams& g, const FillXY f, double* __restrict__ U, unsigned t, int k) {
  const int gw = g.gw, nx = g.nx, ny = g.ny;
  const unsigned full = 2u * gw * (unsigned)g.isize;
  int i, j;
  if (t < full) { const int r = (int)(t / (unsigned)g.isize); i = (int)(t % (unsigned)g.isize); j = r < gw ? r : ny + r; }
  else { const unsigned q = t - full; const int a = (int)(q % (2u * gw)); j = gw + (int)(q / (2u * gw)); i = a < gw ? a : nx + a; }
  const size_t N = g.ncell;
  const size_t krow = (size_t)g.sk * k;
  const bool xg = i < gw || i >= nx + gw, yg = j < gw || j >= ny + gw;
  const int xside = i < gw ? 0 : 1, yside = j < gw ? 0 : 1;
  const int bx = xside ? f.bx1 : f.bx0, by = yside ? f.by1 : f.by0;
  const size_t o_out = krow + (size_t)g.sj * j + i;
  if (f.shear) {
    const int jw = wrap_row(j, gw, ny);   // y periodic
    if (!xg) {   // y ghost of an interior column: the periodic image
      const size_t o_in = krow + (size_t)g.sj * jw + i;
      for (int v = 0; v < 8; ++v) U[o_out + v * N] = U[o_in + v * N];
      return;
    }
    const int il = xside ? i - nx - gw : i;
    for (int v = 0; v < 8; ++v) {
      if (v == IA && xside == 1 && il == 0) {   // the first outer Bx ghost is an evolved face value, kept (MHDRunGodunov.cpp:3727-3735): its y images copy it
        if (yg) U[o_out + v * N] = U[krow + (size_t)g.sj * jw + i + v * N];
        continue;
      }
      U[o_out + v * N] = shear_ghost_value(g, f.sg, U, xside, il, jw, krow, v);
    }
    return;
  }
  const int si = xg ? fill_src_index(bx, xside, nx, gw, i) : i;
  const int sj_ = yg ? fill_src_index(by, yside, ny, gw, j) : j;
  const size_t o_in = krow + (size_t)g.sj * sj_ + si;
  for (int v = 0; v < g.nvar; ++v) {
    double x = U[o_in + v * N];
    if (xg && bx == 1 && v == IU) x = x * -1.0;   // X pass (the order of the reference's passes; both products are exact)
    if (yg && by == 1 && v == IV) x = x * -1.0;   // Y pass
    U[o_out + v * N] = x;
  }
}

// ---- CFL scan: value of one cell, 0 outside the interior (all contributions are >= 0) ----------------------
template <int NV>
RG_DEVFN double hydro_invdt_cell(const DevParams& g, const double* __restrict__ U, unsigned idx) {
  const IJK c = unflatten(g, idx);
  const int gw = g.gw;
  if (c.i < gw || c.i >= g.isize - gw || c.j < gw || c.j >= g.jsize - gw) return 0.0;
  if (NV == 5 && (c.k < gw || c.k >= g.ksize - gw)) return 0.0;
  const size_t N = g.ncell;
  double u[NV], q[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) u[v] = U[idx + v * N];
  const double cs = hydro_prim<NV>(g, u, q);
  const double vx = cs + fabs(q[IU]), vy = cs + fabs(q[IV]);
  if (NV == 5) {
    const double vz = cs + fabs(q[IW]);
    return vx / g.dx + vy / g.dy + vz / g.dz;
  }
  return vx / g.dx + vy / g.dy;
}

RG_DEVFN double mhd_invdt_cell(const DevParams& g, const double* __restrict__ U, unsigned idx) {
  const IJK c = unflatten(g, idx);
  const int gw = g.gw;
  if (c.i < gw || c.i >= g.isize - gw || c.j < gw || c.j >= g.jsize - gw) return 0.0;
  if (g.three_d && (c.k < gw || c.k >= g.ksize - gw)) return 0.0;
  const size_t N = g.ncell;
  double u[8];
#pragma unroll
  for (int v = 0; v < 8; ++v) u[v] = U[idx + v * N];
  const double bnx = U[idx + 1 + IA * N], bny = U[idx + g.sj + IB * N];
  const double bnz = g.three_d ? U[idx + g.sk + IC * N] : 0.0;
  const Prim8 q = mhd_prim(g, u, bnx, bny, bnz, 0.0);
  double sx, sy, sz;
  info_speeds(g, q, sx, sy, sz);
  if (!g.three_d) return sx / g.dx + sy / g.dy;
  if (g.Omega0 > 0) sy += 1.5 * g.Omega0 * g.deltaX / 2;  // shear velocity at the box edge (MHDRunBase.cpp:223-225)
  return sx / g.dx + sy / g.dy + sz / g.dz;
}

// ------------------------------------------------------------------------------------------------------------
// history diagnostics of the MHD runs (MHDRunBase::history_mri / history_default, MHDRunBase.cpp:3311-3407,
// 3476-3619).  The reference copies the whole state to the host and loops over it; here the y sums are done on the
// device, one thread per (i,k) row in increasing j (coalesced along i), a second kernel adds the rows of a column in
// increasing k, and the host adds the isize column values: a fixed summation order, reproducible run to run (it is
// not the reference's k,j,i order, so sums agree to round-off, not bit for bit).
//   q = 0 rho, 1 vx = mx/rho, 2 vy = my/rho            (all i, ghosts included: the y-z means of history_mri)
//       3 magp terms, 4 maxwell term, 5 Bx, 6 By, 7 Bz, 8 divB   (interior i only)
enum { HIST_NQ = 9 };
RG_DEVFN void hist_row_cell(const DevParams& g, const double* __restrict__ U, double* __restrict__ rows, unsigned idx) {
  const int nk = g.three_d ? g.nz : 1;
  const int i = (int)(idx % (unsigned)g.isize), kk = (int)(idx / (unsigned)g.isize);
  if (kk >= nk) return;
  const int k = g.three_d ? kk + g.gw : 0;
  const size_t N = g.ncell;
  const unsigned sj = g.sj, sk = g.three_d ? g.sk : 0u;
  const bool inner = i >= g.gw && i < g.isize - g.gw;
  double acc[HIST_NQ];
#pragma unroll
  for (int q = 0; q < HIST_NQ; ++q) acc[q] = 0.0;
  for (int j = g.gw; j < g.jsize - g.gw; ++j) {
    const unsigned o = (unsigned)i + sj * (unsigned)j + g.sk * (unsigned)k;
    const double rho = U[o + ID * N];
    acc[0] += rho;
    acc[1] += U[o + IU * N] / rho;
    acc[2] += U[o + IV * N] / rho;
    if (inner) {
      const double bx = U[o + IA * N], by = U[o + IB * N], bz = U[o + IC * N];
      const double sx = bx + U[o + 1 + IA * N], sy = by + U[o + sj + IB * N];
      acc[3] += 0.25 * (sx * sx);
      acc[3] += 0.25 * (sy * sy);
      double dv = (U[o + 1 + IA * N] - bx) / g.dx + (U[o + sj + IB * N] - by) / g.dy;
      if (g.three_d) {
        const double sz = bz + U[o + sk + IC * N];
        acc[3] += 0.25 * (sz * sz);
        dv = dv + (U[o + sk + IC * N] - bz) / g.dz;
      }
      acc[4] -= 0.25 * sx * sy;
      acc[5] += bx; acc[6] += by; acc[7] += bz;
      acc[8] += dv;
    }
  }
  const size_t R = (size_t)g.isize * nk;
#pragma unroll
  for (int q = 0; q < HIST_NQ; ++q) rows[(size_t)q * R + idx] = acc[q];
}

// history_turbulence (MHDRunBase.cpp:3626-3810): row sums over the interior y of the 19 quantities of its two loops, interior
// i only.  q = 0 mass, 1 eKin, 2 mean_v2, 3 eMag, 4 helicity, 5-7 mean_B, 8-10 mean_rhov, 11-16 the high-k DFT coefficients of
// Bx (x re/im, y re/im, z re/im; phases from the ghost-inclusive integer indices like the reference), 17 divB, 18 unused
enum { HIST_TURB_NQ = 18 };
RG_DEVFN void hist_turb_row_cell(const DevParams& g, const double* __restrict__ U, double* __restrict__ rows, unsigned idx) {
  const int nk = g.nz;
  const int i = (int)(idx % (unsigned)g.isize), kk = (int)(idx / (unsigned)g.isize);
  if (kk >= nk) return;
  const int k = kk + g.gw;
  const size_t N = g.ncell;
  const unsigned sj = g.sj, sk = g.sk;
  double acc[HIST_TURB_NQ];
#pragma unroll
  for (int q = 0; q < HIST_TURB_NQ; ++q) acc[q] = 0.0;
  if (i >= g.gw && i < g.isize - g.gw) {
    const double pi = 2 * asin(1.0);
    const int kfft = g.nx - 3;
    for (int j = g.gw; j < g.jsize - g.gw; ++j) {
      const unsigned o = (unsigned)i + sj * (unsigned)j + sk * (unsigned)k;
      const double rho = U[o + ID * N];
      const double mu = U[o + IU * N], mv = U[o + IV * N], mw = U[o + IW * N];
      const double bx = U[o + IA * N], by = U[o + IB * N], bz = U[o + IC * N];
      acc[0] += rho;
      acc[1] += (mu * mu) / rho; acc[1] += (mv * mv) / rho; acc[1] += (mw * mw) / rho;
      acc[2] += (mu / rho) * (mu / rho); acc[2] += (mv / rho) * (mv / rho); acc[2] += (mw / rho) * (mw / rho);
      acc[3] += bx * bx; acc[3] += by * by; acc[3] += bz * bz;
      acc[4] += mu * bx / sqrt(rho); acc[4] += mv * by / sqrt(rho); acc[4] += mw * bz / sqrt(rho);
      acc[5] += bx; acc[6] += by; acc[7] += bz;
      acc[8] += mu; acc[9] += mv; acc[10] += mw;
      acc[11] += bx * cos(2 * pi * kfft * i / g.nx); acc[12] += bx * sin(2 * pi * kfft * i / g.nx);
      acc[13] += bx * cos(2 * pi * kfft * j / g.ny); acc[14] += bx * sin(2 * pi * kfft * j / g.ny);
      acc[15] += bx * cos(2 * pi * kfft * k / g.nz); acc[16] += bx * sin(2 * pi * kfft * k / g.nz);
      acc[17] += (U[o + 1 + IA * N] - bx) / g.dx + (U[o + sj + IB * N] - by) / g.dy + (U[o + sk + IC * N] - bz) / g.dz;
    }
  }
  const size_t R = (size_t)g.isize * nk;
#pragma unroll
  for (int q = 0; q < HIST_TURB_NQ; ++q) rows[(size_t)q * R + idx] = acc[q];
}

// random forcing (problem "turbulence"): row sums over the interior y of  rho v.f  and  rho f.f  (the two sums the
// normalisation needs, HydroRunBase.cpp:1229-1243), same rows / columns layout as the history sums, interior i only
RG_DEVFN void forcing_row_cell(const DevParams& g, const double* __restrict__ U, const double* __restrict__ Frc,
                               double* __restrict__ rows, unsigned idx) {
  const int nk = g.nz;
  const int i = (int)(idx % (unsigned)g.isize), kk = (int)(idx / (unsigned)g.isize);
  if (kk >= nk) return;
  const int k = kk + g.gw;
  const size_t N = g.ncell;
  double a0 = 0.0, a1 = 0.0;
  if (i >= g.gw && i < g.isize - g.gw)
    for (int j = g.gw; j < g.jsize - g.gw; ++j) {
      const unsigned o = (unsigned)i + g.sj * (unsigned)j + g.sk * (unsigned)k;
      const double rho = U[o + ID * N];
      const double u = U[o + IU * N] / rho, v = U[o + IV * N] / rho, w = U[o + IW * N] / rho;
      const double uu = Frc[o], vv = Frc[o + N], ww = Frc[o + 2 * N];
      a0 += rho * (u * uu + v * vv + w * ww);
      a1 += rho * uu * uu;
      a1 += rho * vv * vv;
      a1 += rho * ww * ww;
    }
  const size_t R = (size_t)g.isize * nk;
  rows[idx] = a0;
  rows[R + idx] = a1;
}

// add_random_forcing (HydroRunBase.cpp:1397-1428): energy first (with the momenta before the kick), then the momenta
RG_DEVFN void add_forcing_cell(const DevParams& g, double* __restrict__ U, const double* __restrict__ Frc, double norm,
                               unsigned idx) {
  const IJK c = unflatten(g, idx);
  if (c.i < g.gw || c.i >= g.isize - g.gw || c.j < g.gw || c.j >= g.jsize - g.gw || c.k < g.gw || c.k >= g.ksize - g.gw) return;
  const size_t N = g.ncell;
  const double rho = U[idx + ID * N];
  const double fx = Frc[idx], fy = Frc[idx + N], fz = Frc[idx + 2 * N];
  double e = U[idx + IP * N];
  e += U[idx + IU * N] / rho * fx * norm + 0.5 * ((fx * norm) * (fx * norm));
  e += U[idx + IV * N] / rho * fy * norm + 0.5 * ((fy * norm) * (fy * norm));
  e += U[idx + IW * N] / rho * fz * norm + 0.5 * ((fz * norm) * (fz * norm));
  U[idx + IP * N] = e;
  U[idx + IU * N] += rho * fx * norm;
  U[idx + IV * N] += rho * fy * norm;
  U[idx + IW * N] += rho * fz * norm;
}

// ForcingOrnsteinUhlenbeck::add_forcing_field, steps 2 and 3 (Forcing_OrnsteinUhlenbeck.cpp:621-683): the real-space sum of
// the 31 modes at the cell centre accelerates the gas; the internal energy is kept.  kz0 = first global plane of this slab.
RG_DEVFN void ou_forcing_cell(const DevParams& g, double* __restrict__ U, const rgpu_ou::OuModes& M, double dt, double yMin,
                              double zMin, int kz0, unsigned idx) {
  const IJK c = unflatten(g, idx);
  if (c.i < g.gw || c.i >= g.isize - g.gw || c.j < g.gw || c.j >= g.jsize - g.gw || c.k < g.gw || c.k >= g.ksize - g.gw) return;
  const size_t N = g.ncell;
  const double twoPi = 2 * 3.14159265358979323846;
  const double xPos = g.xMin + g.dx / 2 + (c.i - g.gw) * g.dx;
  const double yPos = yMin + g.dy / 2 + (c.j - g.gw) * g.dy;
  const double zPos = zMin + g.dz / 2 + (c.k - g.gw + kz0) * g.dz;
  double A[3] = {0.0, 0.0, 0.0};
  for (int im = 0; im < rgpu_ou::NMODE; ++im) {
    const double phase = xPos * M.mode[im] + yPos * M.mode[rgpu_ou::NMODE + im] + zPos * M.mode[2 * rgpu_ou::NMODE + im];
    const double cs = cos(twoPi * phase);
    A[0] += M.force[im] * cs;
    A[1] += M.force[rgpu_ou::NMODE + im] * cs;
    A[2] += M.force[2 * rgpu_ou::NMODE + im] * cs;
  }
  const double rho = U[idx + ID * N];
  double mu = U[idx + IU * N], mv = U[idx + IV * N], mw = U[idx + IW * N];
  double eInt = 0.5 * (mu * mu + mv * mv + mw * mw) / rho;
  eInt = U[idx + IP * N] - eInt;
  mu += A[0] * dt * rho;
  mv += A[1] * dt * rho;
  mw += A[2] * dt * rho;
  U[idx + IU * N] = mu; U[idx + IV * N] = mv; U[idx + IW * N] = mw;
  U[idx + IP * N] = eInt + 0.5 * (mu * mu + mv * mv + mw * mw) / rho;
}

// state checksum (rgpu_state_checksum): one thread per (i, k) row of the interior adds the 64-bit patterns of all variables of its
// ny cells, modulo 2^64 -- an order-independent sum, so slabs, tiles and launch geometry cannot change it
RG_DEVFN void checksum_row_cell(const DevParams& g, const double* __restrict__ U, unsigned long long* __restrict__ rows, unsigned idx) {
  const int nk = g.three_d ? g.nz : 1;
  const int ii = (int)(idx % (unsigned)g.nx), kk = (int)(idx / (unsigned)g.nx);
  if (kk >= nk) return;
  const int i = ii + g.gw, k = g.three_d ? kk + g.gw : 0;
  const size_t N = g.ncell;
  unsigned long long acc = 0ull;
  for (int j = g.gw; j < g.jsize - g.gw; ++j) {
    const size_t o = (size_t)i + (size_t)g.sj * j + (size_t)g.sk * k;
    for (int v = 0; v < g.nvar; ++v) {
      unsigned long long b;
      const double x = U[o + v * N];
      __builtin_memcpy(&b, &x, sizeof(b));
      acc += b;
    }
  }
  rows[idx] = acc;
}

// column sums: thread (i,q) adds rows[q][k][i] over k
RG_DEVFN void hist_col_cell(const DevParams& g, const double* __restrict__ rows, double* __restrict__ cols, int nq, unsigned idx) {
  const int nk = g.three_d ? g.nz : 1;
  const int i = (int)(idx % (unsigned)g.isize), q = (int)(idx / (unsigned)g.isize);
  if (q >= nq) return;
  const size_t R = (size_t)g.isize * nk;
  double a = 0.0;
  for (int kk = 0; kk < nk; ++kk) a += rows[(size_t)q * R + (size_t)kk * g.isize + i];
  cols[(size_t)q * g.isize + i] = a;
}

// Reynolds stress rows: sum_j rho * dTau * (vx - <vx>(i)) * (vy - <vy>(i)) for interior i (MHDRunBase.cpp:3589-3595)
RG_DEVFN void hist_reynolds_cell(const DevParams& g, const double* __restrict__ U, const double* __restrict__ mean_vx,
                                 const double* __restrict__ mean_vy, double dTau, double* __restrict__ rows, unsigned idx) {
  const int nk = g.three_d ? g.nz : 1;
  const int i = (int)(idx % (unsigned)g.isize), kk = (int)(idx / (unsigned)g.isize);
  if (kk >= nk) return;
  const int k = g.three_d ? kk + g.gw : 0;
  const size_t N = g.ncell;
  double a = 0.0;
  if (i >= g.gw && i < g.isize - g.gw) {
    const double m1 = mean_vx[i], m2 = mean_vy[i];
    for (int j = g.gw; j < g.jsize - g.gw; ++j) {
      const unsigned o = (unsigned)i + g.sj * (unsigned)j + g.sk * (unsigned)k;
      const double rho = U[o + ID * N];
      a += rho * dTau * (U[o + IU * N] / rho - m1) * (U[o + IV * N] / rho - m2);
    }
  }
  rows[idx] = a;
}

}  // namespace rgpu_dev
