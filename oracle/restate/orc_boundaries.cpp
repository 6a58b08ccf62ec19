// orc_boundaries.cpp -- ORACLE (test infrastructure).  Ghost-cell fill.
//   make_boundary2<bct,loc> CPU version      make_boundary_base.h:1040-1332
//   make_boundaries / make_all_boundaries    HydroRunBase.cpp:2276-2342
//   make_jet                                 HydroRunBase.cpp:2374-2408
//   make_boundary2_z_stratified_cpu          make_boundary_base.h:1356-1647 (dispatch HydroRunBase.cpp:2204-2216)
//   make_boundaries_shear / _all_..._shear   MHDRunGodunov.cpp:3539-3793
#include "orc_common.h"

namespace orc {

namespace {

// one face; dir 0,1,2 ; side 0 = min, 1 = max
void fill_face(const Ctx& c, double* U, int dir, int side, int bct) {
  if (bct != RGPU_BC_DIRICHLET && bct != RGPU_BC_NEUMANN && bct != RGPU_BC_PERIODIC) return;  // shear / copy: untouched
  const int gw = c.gw;
  const int n[3] = {c.nx, c.ny, c.nz};
  const int size[3] = {c.isize, c.jsize, c.ksize};
  const int normal_mom = (dir == 0) ? IU : (dir == 1) ? IV : IW;
  for (int v = 0; v < c.nvar; ++v) {
    const int g0 = side == 0 ? 0 : n[dir] + gw;
    for (int gidx = g0; gidx < g0 + gw; ++gidx) {
      double sign = 1.0;
      int src;
      if (bct == RGPU_BC_DIRICHLET) {
        src = side == 0 ? 2 * gw - 1 - gidx : 2 * n[dir] + 2 * gw - 1 - gidx;
        if (v == normal_mom) sign = -1.0;
      } else if (bct == RGPU_BC_NEUMANN) {
        src = side == 0 ? gw : n[dir] + gw - 1;
      } else {
        src = side == 0 ? n[dir] + gidx : gidx - n[dir];
      }
      // the two transverse directions run over their FULL extent (ghosts included)
      const int d1 = (dir + 1) % 3, d2 = (dir + 2) % 3;
      int ijk_out[3], ijk_in[3];
      for (int b = 0; b < size[d2]; ++b)
        for (int a = 0; a < size[d1]; ++a) {
          ijk_out[dir] = gidx; ijk_out[d1] = a; ijk_out[d2] = b;
          ijk_in[dir] = src; ijk_in[d1] = a; ijk_in[d2] = b;
          U[c.idx(ijk_out[0], ijk_out[1], ijk_out[2]) + c.ncell * v] =
              U[c.idx(ijk_in[0], ijk_in[1], ijk_in[2]) + c.ncell * v] * sign;
        }
    }
  }
}

// z faces of the vertically stratified MRI box, ghost width 3, the reference's two loops as written
void fill_face_z_stratified(const Ctx& c, double* U, int side) {
  const rgpu_params& p = c.p;
  const double H = p.cIso / p.Omega0;
  const double dx = c.dx, dy = c.dy, dz = c.dz, zMin = p.zMin, zMax = p.zMax;
  const double factor = -dz / 2.0 / H / H;
  const int imax = c.isize, jmax = c.jsize, kmax = c.ksize;
  const size_t N = c.ncell;
  double ratio_nyp1 = 1.0, ratio_nyp2 = 1.0, ratio_nyp3 = 1.0;
  if (!p.zStratifiedFloor) {
    if (side == 0) {
      ratio_nyp1 = std::exp(factor * (-2 * (zMin + 0.5 * dz) + dz));
      ratio_nyp2 = std::exp(factor * (-2 * (zMin + 0.5 * dz) + 3.0 * dz));
      ratio_nyp3 = std::exp(factor * (-2 * (zMin + 0.5 * dz) + 5.0 * dz));
    } else {
      ratio_nyp1 = std::exp(factor * (2 * (zMax - 0.5 * dz) + dz));
      ratio_nyp2 = std::exp(factor * (2 * (zMax - 0.5 * dz) + 3.0 * dz));
      ratio_nyp3 = std::exp(factor * (2 * (zMax - 0.5 * dz) + 5.0 * dz));
    }
  }
#define UU(i, j, k, v) U[c.idx((i), (j), (k)) + N * (v)]
  if (side == 0) {
    for (int j = 0; j < jmax; j++)
      for (int i = 0; i < imax; i++) {
        const double rho3 = UU(i, j, 3, ID);
        const double rho2 = UU(i, j, 3, ID) * ratio_nyp1;
        const double rho1 = UU(i, j, 3, ID) * ratio_nyp1 * ratio_nyp2;
        const double rho0 = UU(i, j, 3, ID) * ratio_nyp1 * ratio_nyp2 * ratio_nyp3;
        UU(i, j, 2, ID) = rho2; UU(i, j, 1, ID) = rho1; UU(i, j, 0, ID) = rho0;
        for (int v = IU; v <= IV; ++v) {   // tangential velocities kept
          UU(i, j, 0, v) = UU(i, j, 3, v) / rho3 * rho0;
          UU(i, j, 1, v) = UU(i, j, 3, v) / rho3 * rho1;
          UU(i, j, 2, v) = UU(i, j, 3, v) / rho3 * rho2;
        }
        const double w = std::fmin(UU(i, j, 3, IW), 0.0);
        UU(i, j, 0, IW) = w; UU(i, j, 1, IW) = w; UU(i, j, 2, IW) = w;
        for (int k = 0; k < 3; ++k) { UU(i, j, k, IA) = 0.0; UU(i, j, k, IB) = 0.0; }
      }
    for (int j = 0; j < jmax - 1; j++)
      for (int i = 0; i < imax - 1; i++) {
        double dbxdx = (UU(i + 1, j, 2, IA) - UU(i, j, 2, IA)) / dx;
        double dbydy = (UU(i, j + 1, 2, IB) - UU(i, j, 2, IB)) / dy;
        const double bz = UU(i, j, 3, IC);
        const double dbz2 = dz * (dbxdx + dbydy);
        UU(i, j, 2, IC) = bz + dbz2;
        dbxdx = (UU(i + 1, j, 1, IA) - UU(i, j, 1, IA)) / dx;
        dbydy = (UU(i, j + 1, 1, IB) - UU(i, j, 1, IB)) / dy;
        const double dbz1 = dz * (dbxdx + dbydy);
        UU(i, j, 1, IC) = bz + dbz2 + dbz1;
        dbxdx = (UU(i + 1, j, 0, IA) - UU(i, j, 0, IA)) / dx;
        dbydy = (UU(i, j + 1, 0, IB) - UU(i, j, 0, IB)) / dy;
        const double dbz0 = dz * (dbxdx + dbydy);
        UU(i, j, 0, IC) = bz + dbz2 + dbz1 + dbz0;
      }
  } else {
    for (int j = 0; j < jmax; j++)
      for (int i = 0; i < imax; i++) {
        const double rho4 = UU(i, j, kmax - 4, ID);
        const double rho3 = UU(i, j, kmax - 4, ID) * ratio_nyp1;
        const double rho2 = UU(i, j, kmax - 4, ID) * ratio_nyp1 * ratio_nyp2;
        const double rho1 = UU(i, j, kmax - 4, ID) * ratio_nyp1 * ratio_nyp2 * ratio_nyp3;
        UU(i, j, kmax - 3, ID) = rho3; UU(i, j, kmax - 2, ID) = rho2; UU(i, j, kmax - 1, ID) = rho1;
        for (int v = IU; v <= IV; ++v) {
          UU(i, j, kmax - 3, v) = UU(i, j, kmax - 4, v) / rho4 * rho3;
          UU(i, j, kmax - 2, v) = UU(i, j, kmax - 4, v) / rho4 * rho2;
          UU(i, j, kmax - 1, v) = UU(i, j, kmax - 4, v) / rho4 * rho1;
        }
        const double w = std::fmax(UU(i, j, kmax - 4, IW), 0.0);
        UU(i, j, kmax - 3, IW) = w; UU(i, j, kmax - 2, IW) = w; UU(i, j, kmax - 1, IW) = w;
        for (int k = kmax - 3; k < kmax; ++k) { UU(i, j, k, IA) = 0.0; UU(i, j, k, IB) = 0.0; }
      }
    for (int j = 0; j < jmax - 1; j++)
      for (int i = 0; i < imax - 1; i++) {
        // (the reference differences By with itself here: U[offset] - U[offset])
        double dbxdx = (UU(i + 1, j, kmax - 3, IA) - UU(i, j, kmax - 3, IA)) / dx;
        double dbydy = (UU(i, j, kmax - 3, IB) - UU(i, j, kmax - 3, IB)) / dy;
        const double bz = UU(i, j, kmax - 3, IC);
        const double dbz1 = dz * (dbxdx + dbydy);
        UU(i, j, kmax - 2, IC) = bz - dbz1;
        dbxdx = (UU(i + 1, j, kmax - 2, IA) - UU(i, j, kmax - 2, IA)) / dx;
        dbydy = (UU(i, j, kmax - 2, IB) - UU(i, j, kmax - 2, IB)) / dy;
        const double dbz2 = dz * (dbxdx + dbydy);
        UU(i, j, kmax - 1, IC) = bz - dbz1 - dbz2;
      }
  }
#undef UU
}

void make_jet(const Ctx& c, double* U) {
  const rgpu_params& p = c.p;
  const int gw = c.gw;
  if (!c.three_d) {
    for (int j = 0; j < gw; j++)
      for (int i = gw + p.offsetJet; i < gw + p.offsetJet + p.ijet; i++) {
        U[c.idx(i, j, 0) + c.ncell * ID] = p.djet;
        U[c.idx(i, j, 0) + c.ncell * IP] = p.pjet / (p.gamma0 - 1.) + 0.5 * p.djet * p.ujet * p.ujet;
        U[c.idx(i, j, 0) + c.ncell * IU] = 0.0;
        U[c.idx(i, j, 0) + c.ncell * IV] = p.djet * p.ujet;
      }
  } else {
    for (int k = 0; k < gw; ++k)
      for (int j = gw + p.offsetJet; j < gw + p.offsetJet + p.ijet; ++j)
        for (int i = gw + p.offsetJet; i < gw + p.offsetJet + p.ijet; ++i) {
          U[c.idx(i, j, k) + c.ncell * ID] = p.djet;
          U[c.idx(i, j, k) + c.ncell * IP] = p.pjet / (p.gamma0 - 1.) + 0.5 * p.djet * p.ujet * p.ujet;
          U[c.idx(i, j, k) + c.ncell * IU] = 0.0;
          U[c.idx(i, j, k) + c.ncell * IV] = 0.0;
          U[c.idx(i, j, k) + c.ncell * IW] = p.djet * p.ujet;
        }
  }
}

}  // namespace

void make_boundaries(const Ctx& c, double* U, int idim) {
  const int dir = idim - 1;
  if (!c.three_d && dir == 2) return;
  fill_face(c, U, dir, 0, c.p.bc[2 * dir]);
  fill_face(c, U, dir, 1, c.p.bc[2 * dir + 1]);
  if (dir == 2 && c.p.bc[4] == RGPU_BC_Z_STRATIFIED) fill_face_z_stratified(c, U, 0);
  if (dir == 2 && c.p.bc[5] == RGPU_BC_Z_STRATIFIED) fill_face_z_stratified(c, U, 1);
  // the jet is re-imposed after the Y fill in 2D and after the Z fill in 3D
  if (c.p.enableJet && ((!c.three_d && dir == 1) || (c.three_d && dir == 2))) make_jet(c, U);
}

void make_boundaries_shear(const Ctx& c, double* Ud, double totalTime, double dt) {
  const rgpu_params& p = c.p;
  const int gw = c.gw, nx = c.nx, ny = c.ny, isize = c.isize, jsize = c.jsize, ksize = c.ksize, nbVar = c.nvar;
  (void)isize;
  Field U; U.wrap(c, Ud, nbVar);
  double deltay, epsi, eps, lambda;
  int jplus, jremap, jremapp1;
  deltay = 1.5 * p.Omega0 * (p.dx * p.nx) * (totalTime + dt);
  deltay = fmod(deltay, (p.dy * p.ny));
  jplus = (int)(deltay / c.dy);
  epsi = fmod(deltay, c.dy);

  // border copies: the gw innermost INTERIOR columns on each side (shearBorderUtils.h:46-112)
  const size_t bsz = (size_t)gw * jsize * ksize;
  std::vector<double> bmin(bsz * nbVar), bmax(bsz * nbVar), smin(bsz * nbVar, 0.0), smax(bsz * nbVar, 0.0);
  auto B = [&](std::vector<double>& b, int i, int j, int k, int v) -> double& {
    return b[(size_t)i + (size_t)gw * (j + (size_t)jsize * k) + bsz * v];
  };
  for (int v = 0; v < nbVar; ++v)
    for (int k = 0; k < ksize; ++k)
      for (int j = 0; j < jsize; ++j)
        for (int i = 0; i < gw; ++i) {
          B(bmin, i, j, k, v) = U(gw + i, j, k, v);
          B(bmax, i, j, k, v) = U(nx + i, j, k, v);
        }

  const double slope_type = p.slope_type;
  if (slope_type == 1 || slope_type == 2) {
    double dsgn, dlim, dcen, dlft, drgt, slop;
    for (int k = 0; k < ksize; k++)
      for (int j = 1; j < jsize - 1; j++)
        for (int i = 0; i < gw; i++)
          for (int iVar = 0; iVar < nbVar; iVar++) {
            if (iVar == IB) {
              B(smin, i, j, k, IB) = B(bmin, i, j + 1, k, IB) - B(bmin, i, j, k, IB);
              B(smax, i, j, k, IB) = B(bmax, i, j + 1, k, IB) - B(bmax, i, j, k, IB);
            } else {
              dlft = slope_type * (B(bmin, i, j, k, iVar) - B(bmin, i, j - 1, k, iVar));
              drgt = slope_type * (B(bmin, i, j + 1, k, iVar) - B(bmin, i, j, k, iVar));
              dcen = 0.5 * (dlft + drgt) / slope_type;
              dsgn = (dcen >= 0.0) ? 1.0 : -1.0;
              slop = fmin(fabs(dlft), fabs(drgt));
              dlim = slop;
              if ((dlft * drgt) <= 0.0) dlim = 0.0;
              B(smin, i, j, k, iVar) = dsgn * fmin(dlim, fabs(dcen));

              dlft = slope_type * (B(bmax, i, j, k, iVar) - B(bmax, i, j - 1, k, iVar));
              drgt = slope_type * (B(bmax, i, j + 1, k, iVar) - B(bmax, i, j, k, iVar));
              dcen = 0.5 * (dlft + drgt) / slope_type;
              dsgn = (dcen >= 0.0) ? 1.0 : -1.0;
              slop = fmin(fabs(dlft), fabs(drgt));
              dlim = slop;
              if ((dlft * drgt) <= 0.0) dlim = 0.0;
              B(smax, i, j, k, iVar) = dsgn * fmin(dlim, fabs(dcen));
            }
          }
  }

  for (int k = 0; k < ksize; k++)
    for (int j = gw; j < jsize - gw; j++) {
      // inner (XMIN) border
      jremap = j - jplus - 1;
      jremapp1 = jremap + 1;
      eps = 1.0 - epsi / c.dy;
      if (jremap < gw) jremap += ny;
      if (jremapp1 < gw) jremapp1 += ny;
      lambda = 0.5 * eps * (eps - 1.0);
      for (int iVar = 0; iVar < nbVar; iVar++)
        for (int i = 0; i < gw; i++) {
          if (iVar == IB) {
            U(i, j, k, IB) = B(bmax, i, jremap, k, IB) + eps * B(smax, i, jremap, k, IB);
          } else {
            U(i, j, k, iVar) = (1.0 - eps) * B(bmax, i, jremap, k, iVar) + eps * B(bmax, i, jremapp1, k, iVar) +
                               lambda * (B(smax, i, jremap, k, iVar) - B(smax, i, jremapp1, k, iVar));
          }
        }
      // outer (XMAX) border
      jremap = j + jplus;
      jremapp1 = jremap + 1;
      eps = epsi / c.dy;
      if (jremap > ny + gw - 1) jremap -= ny;
      if (jremapp1 > ny + gw - 1) jremapp1 -= ny;
      lambda = 0.5 * eps * (eps - 1.0);
      for (int iVar = 0; iVar < nbVar; iVar++)
        for (int i = 0; i < gw; i++) {
          if (iVar < 5) {
            U(nx + gw + i, j, k, iVar) = (1.0 - eps) * B(bmin, i, jremap, k, iVar) + eps * B(bmin, i, jremapp1, k, iVar) +
                                         lambda * (B(smin, i, jremapp1, k, iVar) - B(smin, i, jremap, k, iVar));
          }
          if (iVar == IA) {
            if (i > 0) {  // the first outer Bx ghost is an evolved face: never overwritten
              U(nx + gw + i, j, k, IA) = (1.0 - eps) * B(bmin, i, jremap, k, IA) + eps * B(bmin, i, jremapp1, k, IA) +
                                         lambda * (B(smin, i, jremapp1, k, IA) - B(smin, i, jremap, k, IA));
            }
          }
          if (iVar == IB) {
            U(nx + gw + i, j, k, IB) = B(bmin, i, jremap, k, IB) + eps * B(smin, i, jremap, k, IB);
          }
          if (iVar == IC) {
            U(nx + gw + i, j, k, IC) = (1.0 - eps) * B(bmin, i, jremap, k, IC) + eps * B(bmin, i, jremapp1, k, IC) +
                                       lambda * (B(smin, i, jremapp1, k, IC) - B(smin, i, jremap, k, IC));
          }
        }
    }
}

void make_all_boundaries(const Ctx& c, double* U, double totalTime, double dt) {
  if (c.p.shearingBoxEnabled && c.three_d) {
    make_boundaries(c, U, RGPU_YDIR);
    make_boundaries_shear(c, U, totalTime, dt);
    make_boundaries(c, U, RGPU_ZDIR);
    make_boundaries(c, U, RGPU_YDIR);
  } else {
    make_boundaries(c, U, RGPU_XDIR);
    make_boundaries(c, U, RGPU_YDIR);
    if (c.three_d) make_boundaries(c, U, RGPU_ZDIR);
  }
}

}  // namespace orc
