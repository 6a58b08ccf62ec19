// api/boundaries.h -- ghost fill: per face, the jet, the shearing box, the one-launch in-plane fill (kernels_bc.h).  See api/ctx.h.
#pragma once
namespace {
// ---- boundaries -----------------------------------------------------------------------------------------------
// x and y faces are indexed with k slowest, so planes [k_lo,k_hi) of a face are one contiguous index range
int launch_face(rgpu_ctx* c, double* U, int dir, int side, int k_lo, int k_hi) {
  const int bct = c->p.bc[2 * dir + side];
  if (bct == RGPU_BC_Z_STRATIFIED && dir == 2) {
    // hydrostatic density ratios of the three ghost planes (make_boundary_base.h:1366-1397), host exp() like the reference
    const rgpu_params& p = c->p;
    const double H = p.cIso / p.Omega0;
    const double factor = -p.dz / 2.0 / H / H;
    ZStrat zs = {1.0, 1.0, 1.0};
    if (!p.zStratifiedFloor) {
      if (side == 0) {
        zs.r1 = std::exp(factor * (-2 * (p.zMin + 0.5 * p.dz) + p.dz));
        zs.r2 = std::exp(factor * (-2 * (p.zMin + 0.5 * p.dz) + 3.0 * p.dz));
        zs.r3 = std::exp(factor * (-2 * (p.zMin + 0.5 * p.dz) + 5.0 * p.dz));
      } else {
        zs.r1 = std::exp(factor * (2 * (p.zMax - 0.5 * p.dz) + p.dz));
        zs.r2 = std::exp(factor * (2 * (p.zMax - 0.5 * p.dz) + 3.0 * p.dz));
        zs.r3 = std::exp(factor * (2 * (p.zMax - 0.5 * p.dz) + 5.0 * p.dz));
      }
    }
    K_bc_zstrat k = {c->g, zs, U, side, stop_clk(c)};
    return rg_launch<kBlock>(c->stream, (unsigned)c->g.isize * c->g.jsize, k);
  }
  if (bct != RGPU_BC_DIRICHLET && bct != RGPU_BC_NEUMANN && bct != RGPU_BC_PERIODIC) return 0;  // shear / copy: untouched
  const DevParams& g = c->g;
  K_bc_face k = {g, U, dir, side, bct, stop_clk(c)};
  if (dir == 2) return rg_launch<kBlock>(c->stream, (unsigned)g.isize * g.jsize * g.gw, k);
  const unsigned per_plane = (dir == 0) ? (unsigned)g.gw * g.jsize : (unsigned)g.isize * g.gw;
  return rg_launch_range<kBlock>(c->stream, per_plane * (unsigned)k_lo, per_plane * (unsigned)(k_hi - k_lo), k);
}

int launch_jet(rgpu_ctx* c, double* U) {
  const rgpu_params& p = c->p;
  if (!p.enableJet || p.ijet <= 0) return 0;
  JetParams jp;
  jp.ijet = p.ijet; jp.offsetJet = p.offsetJet; jp.djet = p.djet;
  jp.ejet = p.pjet / (p.gamma0 - 1.) + 0.5 * p.djet * p.ujet * p.ujet;   // HydroRunBase.cpp:2383
  jp.mjet = p.djet * p.ujet;
  const unsigned n = c->g.three_d ? (unsigned)p.ijet * p.ijet * c->g.gw : (unsigned)p.ijet * c->g.gw;
  K_jet k = {c->g, jp, U, stop_clk(c)};
  return rg_launch<kBlock>(c->stream, n, k);
}

int do_make_boundaries(rgpu_ctx* c, double* U, int idim, int k_lo = 0, int k_hi = -1) {
  const int dir = idim - 1;
  if (dir < 0 || dir > 2) return -1;
  if (stop_now(c)) return 0;
  if (!c->g.three_d && dir == 2) return 0;
  if (k_hi < 0) k_hi = c->g.ksize;
  {
    // two faces of the same plain kind (mirror / copy / periodic): one launch for both
    const int b0 = c->p.bc[2 * dir], b1 = c->p.bc[2 * dir + 1];
    auto plain = [](int b) { return b == RGPU_BC_DIRICHLET || b == RGPU_BC_NEUMANN || b == RGPU_BC_PERIODIC; };
    if (plain(b0) && plain(b1)) {
      const DevParams& g = c->g;
      K_bc_faces k = {g, U, dir, b0, b1, 0u, stop_clk(c)};
      if (dir == 2) {
        k.n = (unsigned)g.isize * g.jsize * g.gw;
        if (rg_launch<kBlock>(c->stream, 2u * k.n, k)) return -1;
      } else {
        // x and y faces are indexed with k slowest: planes [k_lo,k_hi) of a face are one contiguous index range
        const unsigned per_plane = (dir == 0) ? (unsigned)g.gw * g.jsize : (unsigned)g.isize * g.gw;
        const unsigned first = per_plane * (unsigned)k_lo, cnt = per_plane * (unsigned)(k_hi - k_lo);
        K_bc_faces kr = {g, U, dir, b0, b1, cnt, stop_clk(c)};
        K_bc_faces_range kk = {kr, first};
        if (rg_launch<kBlock>(c->stream, 2u * cnt, kk)) return -1;
      }
    } else if (launch_face(c, U, dir, 0, k_lo, k_hi) || launch_face(c, U, dir, 1, k_lo, k_hi)) return -1;
  }
  // the jet is re-imposed after the Y fill in 2D and after the Z fill in 3D (HydroRunBase.cpp:2286-2312)
  if (c->p.enableJet && ((!c->g.three_d && dir == 1) || (c->g.three_d && dir == 2 && c->p.bc[4] != RGPU_BC_COPY)))
    return launch_jet(c, U);
  return 0;
}

int do_make_boundaries_shear(rgpu_ctx* c, double* U, double totalTime, double dt, int k_lo = 0, int k_hi = -1) {
  const rgpu_params& p = c->p;
  if (c->clk_cur && !RG_SYNC_LAUNCH) return -1;   // (the separate shear pass takes its offsets by value: device-clock steps use the fused fill)
  if (stop_now(c)) return 0;
  // MHDRunGodunov.cpp:3554-3557
  double deltay = 1.5 * p.Omega0 * (p.dx * p.nx) * (totalTime + dt);
  deltay = std::fmod(deltay, (p.dy * p.ny));
  ShearGhost sg;
  sg.jplus = (int)(deltay / p.dy);
  const double epsi = std::fmod(deltay, p.dy);
  sg.eps_min = 1.0 - epsi / p.dy;
  sg.eps_max = epsi / p.dy;
  if (k_hi < 0) k_hi = c->g.ksize;
  const unsigned per_plane = (unsigned)c->g.gw * c->g.ny;
  K_shear_ghost k = {c->g, sg, U};
  return rg_launch_range<kBlock>(c->stream, per_plane * (unsigned)k_lo, per_plane * (unsigned)(k_hi - k_lo), k);
}

// ---- the in-plane ghost fill in one launch (kernels_bc.h: fill_xy_cell) -----------------------------------------------
// x and y faces (and the shearing-box remap of the x borders) act within one z plane and leave, in every ghost cell, a function of
// that plane's interior cells: one thread per ghost cell, one launch for up to two ranges of planes, instead of X, Y (plain) or
// Y, shear, Y (shearing box) per range.  Possible when the x / y faces are plain (mirror / copy / periodic) or the shearing box
// with periodic y; the 2D jet (re-imposed after the Y pass) is launched behind it.
bool fill_xy_plan(const rgpu_ctx* c, double totalTime, double dt, FillXY* f) {
  const rgpu_params& p = c->p;
  auto plain = [](int b) { return b == RGPU_BC_DIRICHLET || b == RGPU_BC_NEUMANN || b == RGPU_BC_PERIODIC; };
  f->bx0 = p.bc[0]; f->bx1 = p.bc[1]; f->by0 = p.bc[2]; f->by1 = p.bc[3]; f->shear = 0;
  f->sg.jplus = 0; f->sg.eps_min = 0.0; f->sg.eps_max = 0.0;
  if (c->g.rot && c->g.shearbox && c->g.three_d) {
    if (p.bc[2] != RGPU_BC_PERIODIC || p.bc[3] != RGPU_BC_PERIODIC) return false;
    double deltay = 1.5 * p.Omega0 * (p.dx * p.nx) * (totalTime + dt);   // MHDRunGodunov.cpp:3554-3557 (do_make_boundaries_shear)
    deltay = std::fmod(deltay, (p.dy * p.ny));
    f->sg.jplus = (int)(deltay / p.dy);
    const double epsi = std::fmod(deltay, p.dy);
    f->sg.eps_min = 1.0 - epsi / p.dy;
    f->sg.eps_max = epsi / p.dy;
    f->shear = 1;
    return true;
  }
  return plain(p.bc[0]) && plain(p.bc[1]) && plain(p.bc[2]) && plain(p.bc[3]);
}
// ... planes [a1, b1) and [a2, b2) of U (either may be empty)
int launch_fill_xy(rgpu_ctx* c, double* U, const FillXY& f, int a1, int b1, int a2, int b2) {
  const int ks = c->g.ksize;
  a1 = a1 < 0 ? 0 : a1; b1 = b1 > ks ? ks : b1; a2 = a2 < 0 ? 0 : a2; b2 = b2 > ks ? ks : b2;
  const int n1 = b1 > a1 ? b1 - a1 : 0, n2 = b2 > a2 ? b2 - a2 : 0;
  if (n1 + n2 == 0) return 0;
  const unsigned per = 2u * (unsigned)c->g.gw * (unsigned)(c->g.isize + c->g.ny);   // ghost cells of one plane (fill_xy_cell)
  if (stop_now(c)) return 0;
  K_fill_xy k = {c->g, f, U, per, a1, n1, a2, (f.shear && !RG_SYNC_LAUNCH) ? c->clk_cur : stop_clk(c)};
  if (rg_launch<kBlock>(c->stream, per * (unsigned)(n1 + n2), k)) return -1;
  if (c->p.enableJet && !c->g.three_d) return launch_jet(c, U);   // 2D: re-imposed after the Y pass (HydroRunBase.cpp:2286-2312)
  return 0;
}
// Z pass of a full fill whose X / Y passes were fused: complete ghost planes come out of complete interior planes when the z faces
// copy planes cell by cell (mirror / copy / periodic / neighbour slab) -- not the stratified face, which treats the last row and
// column of a plane differently
// ... and only a single-domain context knows that about the whole box (another slab of the run may own a stratified face and would
// send planes whose corners still wait for its last Y pass): slab contexts keep the separate passes in the whole-domain pieces;
// their overlapped schedule fills plane ranges (step_fill_planes), which is fused whatever the z faces are
bool z_fill_is_planewise(const rgpu_ctx* c) {
  if (c->p.slab_count > 1) return false;
  // (RGPU_BC_COPY z faces with slab_count == 1 -- the self-ring, or an external z driver -- count as plane-wise: the supplier of
  // the planes must send them complete, x / y ghost cells and corners included; stated in include/rgpu.h at RGPU_BC_COPY)
  return !c->g.three_d || (c->p.bc[4] != RGPU_BC_Z_STRATIFIED && c->p.bc[5] != RGPU_BC_Z_STRATIFIED);
}

}  // namespace
