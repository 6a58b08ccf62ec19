// api/step.h -- the step: pre / core / post pieces, the hydro, 2D MHD and 3D MHD cores over plane ranges, the dissipative stage,
// the CFL scan.  See api/ctx.h.
#pragma once
namespace {
// ---- the step -------------------------------------------------------------------------------------------------
int step_pre(rgpu_ctx* c, int nStep) {
  if (c->g.rot) return 0;
  if (c->ghost_ok_parity == nStep % 2) return 0;   // the kernel that wrote this state filled its ghost cells too (periodic images)
  Phase ph(c, RGPU_T_BOUNDARIES);
  double* in = c->U[nStep % 2];
  FillXY f;
  if (z_fill_is_planewise(c) && fill_xy_plan(c, 0.0, 0.0, &f)) {   // X and Y in one launch over the interior planes, then Z copies whole planes
    if (launch_fill_xy(c, in, f, c->g.three_d ? c->g.gw : 0, c->g.three_d ? c->g.ksize - c->g.gw : 1, 0, 0)) return -1;
  } else if (do_make_boundaries(c, in, RGPU_XDIR) || do_make_boundaries(c, in, RGPU_YDIR)) return -1;
  if (c->g.three_d && do_make_boundaries(c, in, RGPU_ZDIR)) return -1;
  return 0;
}

int step_post_a(rgpu_ctx* c, int nStep, double dt_arg, double t_arg) {
  if (!c->g.rot) return 0;
  const StepTime st = step_time(c, dt_arg, t_arg);
  if (st.skip) return 0;
  const double dt = st.dt, totalTime = st.t;
  Phase ph(c, RGPU_T_BOUNDARIES);
  double* out = c->U[(nStep + 1) % 2];
  FillXY f;
  if (z_fill_is_planewise(c) && fill_xy_plan(c, totalTime, dt, &f))   // Y, shear, [Z], Y (or X, Y) as one pass over the interior planes; post_b adds Z
    return launch_fill_xy(c, out, f, c->g.three_d ? c->g.gw : 0, c->g.three_d ? c->g.ksize - c->g.gw : 1, 0, 0);
  if (c->g.shearbox && c->g.three_d) {
    if (do_make_boundaries(c, out, RGPU_YDIR)) return -1;
    return do_make_boundaries_shear(c, out, totalTime, dt);
  }
  if (do_make_boundaries(c, out, RGPU_XDIR) || do_make_boundaries(c, out, RGPU_YDIR)) return -1;
  return 0;
}

int step_post_b(rgpu_ctx* c, int nStep) {
  if (!c->g.rot) return 0;
  Phase ph(c, RGPU_T_BOUNDARIES);
  double* out = c->U[(nStep + 1) % 2];
  if (c->g.three_d && do_make_boundaries(c, out, RGPU_ZDIR)) return -1;
  FillXY f;
  if (c->g.shearbox && c->g.three_d && !(z_fill_is_planewise(c) && fill_xy_plan(c, 0.0, 0.0, &f))) return do_make_boundaries(c, out, RGPU_YDIR);
  return 0;   // (fused post_a: the z ghost planes are copies of complete planes, the last Y pass has nothing left to do)
}

// In-plane part of the ghost fill of the step's OUTPUT state, restricted to planes [a,b) (and [a2,b2)): what a z-slab driver applies
// to the planes it is about to send, so that the neighbour receives finished planes (x / y ghosts and corners
// included) and never has to touch its z ghost planes again.  x and y fills (and the shear remap) act within one
// z plane, hence plane-wise { Y, shear, Y } + copying planes equals the reference's { Y, shear, Z, Y } sequence.
int step_fill_planes(rgpu_ctx* c, int nStep, double dt_arg, double t_arg, int a, int b, int a2 = 0, int b2 = 0) {
  const StepTime st = step_time(c, dt_arg, t_arg);
  if (st.skip) return 0;
  const double dt = st.dt, totalTime = st.t;
  Phase ph(c, RGPU_T_BOUNDARIES);
  double* out = c->U[(nStep + 1) % 2];
  FillXY f;
  if (fill_xy_plan(c, totalTime, dt, &f)) return launch_fill_xy(c, out, f, a, b, a2, b2);
  for (int n = 0; n < 2; ++n) {
    const int lo = n ? a2 : a, hi = n ? b2 : b;
    if (hi <= lo) continue;
    if (c->g.rot && c->g.shearbox) {
      if (do_make_boundaries(c, out, RGPU_YDIR, lo, hi) || do_make_boundaries_shear(c, out, totalTime, dt, lo, hi) ||
          do_make_boundaries(c, out, RGPU_YDIR, lo, hi)) return -1;
    } else if (do_make_boundaries(c, out, RGPU_XDIR, lo, hi) || do_make_boundaries(c, out, RGPU_YDIR, lo, hi)) return -1;
  }
  return 0;
}

// ---- plane-range helpers -----------------------------------------------------------------------------------------
// Every kernel body works on a flat cell index and guards its own (i,j,k) validity, so a stage can be run on any
// range of z planes.  The step is expressed as "complete the UPDATE of planes [a,b)"; each stage then has to cover
//   update [a,b) <- flux/emf [a,b+1) <- trace [a-1,b+1) <- elec [a-1,b+2), prim [a-2,b+2)      (3D MHD)
//   update [a,b) <- flux [a,b+1) <- trace [a-1,b+1) <- prim [a-2,b+2)                           (hydro)
// clipped to the array.  Values are deterministic functions of the (unchanging) input state, so computing a plane
// twice in two calls is harmless; a z-slab driver uses this to update the planes that do not depend on the
// neighbours' ghost planes while the halo exchange is still in flight.
struct PlaneRange { int lo, hi; };
inline PlaneRange clip(int lo, int hi, int ksize) {
  PlaneRange r = {lo < 0 ? 0 : lo, hi > ksize ? ksize : hi};
  if (r.hi < r.lo) r.hi = r.lo;
  return r;
}
template <int BLOCK, int MINW, class K>
int launch_planes(rg_stream_t s, const DevParams& g, PlaneRange r, const K& k) {
  if (r.hi <= r.lo) return 0;
  return rg_launch_planes<BLOCK, MINW>(s, (unsigned)r.lo * g.sk, g.sk, (unsigned)(r.hi - r.lo), k, (unsigned)g.xcd_sub);
}

// Can the update kernels of a 3D MHD step carry the CFL scan of the new state (see mhd3d_core)?  Depends on this
// context's boundary types: slabs of one run may answer differently (the slab driver agrees on the minimum once, at
// rgpu_comm_create, through rgpu_inv_dt_fusable).
bool mhd3d_scan_cond(const rgpu_ctx* c) {
  const rgpu_params& p = c->p;
  const DevParams& g = c->g;
  if (g.grav_on == 2 || p.nu > 0 || p.eta > 0 || p.randomForcingEnabled || p.ouForcingEnabled) return false;
  if (g.rot) {
    const bool xy_ok = (p.bc[0] == RGPU_BC_PERIODIC || p.bc[0] == RGPU_BC_SHEARINGBOX) && p.bc[1] == p.bc[0];
    auto zok = [](int b) { return b == RGPU_BC_PERIODIC || b == RGPU_BC_COPY; };
    return xy_ok && p.bc[2] == RGPU_BC_PERIODIC && p.bc[3] == RGPU_BC_PERIODIC && zok(p.bc[4]) && zok(p.bc[5]);
  }
  return true;
}
bool hydro3d_scan_cond(const rgpu_ctx* c) {
  return !(c->p.nu > 0) && !c->p.randomForcingEnabled && !c->p.ouForcingEnabled && rgpu_tiled::hydro3d_sweep_covers(c->g) && c->g.grav_on != 2;
}

// hydro: launch-time specialisation on the Riemann solver and the slope type (launchers.h); the no-gravity instantiations
// only, everything else runs the generic kernels
template <int ND, int NV, int SPEC>
int hydro_flux_trace_spec(rgpu_ctx* c, double dtdx, double dtdy, double dtdz, int a, int b) {
  const DevParams& g = c->g;
  const int ks = g.ksize;
  { Phase ph(c, RGPU_T_TRACE); K_hydro_trace<ND, NV, SPEC> k = {g, c->Q, c->T, dtdx, dtdy, dtdz}; if (launch_planes<kBlock, 1>(c->stream, g, clip(a - 1, b + 1, ks), k)) return -1; }
  { Phase ph(c, RGPU_T_FLUX); K_hydro_flux<ND, NV, false, SPEC> k = {g, c->T, c->F}; if (launch_planes<kBlockHeavy, 1>(c->stream, g, clip(a, b + 1, ks), k)) return -1; }
  return 0;
}

template <int ND, int NV>
// (a2, b2): 3D, tiled sweep only -- a second plane range in the same launch (the two boundary ranges of a slab)
int hydro_core(rgpu_ctx* c, const double* in, double* out, double dt_arg, int a, int b, bool acc_piece = false, int a2 = 0, int b2 = 0) {
  const DevParams& g = c->g;
  const StepTime st = step_time(c, dt_arg, 0.0);
  if (st.skip) return 0;
  const double dt = st.dt;
  const double dtdx = dt / g.dx, dtdy = dt / g.dy, dtdz = dt / g.dz;
  const int ks = g.ksize;
  if (ND == 3) {   // LDS-tiled z-marching sweep: the whole step in one kernel (hip/tiled_hydro.h)
    Phase ph(c, RGPU_T_SWEEP);
    // whole-domain steps whose output nothing modifies afterwards carry the CFL scan of the new state along; slab pieces
    // (acc_piece: RGPU_CORE_UPDATE | RGPU_CORE_SCAN after a reset by the FLUXES call) accumulate into the same slot
    const bool cond = hydro3d_scan_cond(c);
    const bool scan = a <= 0 && b >= ks && cond && !acc_piece;
    const bool piece = acc_piece && cond && c->scan_acc_parity == ((out == c->U[0]) ? 0 : 1);
    if (acc_piece && !piece) c->scan_acc_parity = -1;
    if (scan && !c->clk_cur && rg_memset_async(c->d_red, 0, RG_DT_SLOTS * sizeof(unsigned long long), c->stream)) return -1;   // (a clock kernel zeroed them)
    const int rc = rgpu_tiled::hydro3d_sweep(c->stream, g, in, out, dtdx, dtdy, dtdz, a, b, (scan || piece) ? c->d_red : 0, st.clk, a2, b2);
    if (rc == 0 && scan) { c->fused_dt_parity = (out == c->U[0]) ? 0 : 1; c->fused_dt_slots = 1; }
    if (rc <= 0) return rc;
    if (st.clk) return -1;   // the flat kernels take dt by value
    if (acc_piece) c->scan_acc_parity = -1;   // flat kernels took over: no accumulated scan for this step
  }
  // the CFL scan of the new state rides in the kernel that writes it when the whole domain is updated in this call and nothing
  // modifies the state afterwards (2D: the fused step or the flat update kernel; 3D with a per-cell gravity field: the flat one)
  const bool scan2 = a <= 0 && b >= ks && !(c->p.nu > 0) && !c->p.randomForcingEnabled && !c->p.ouForcingEnabled;
  const bool folding = c->clk_cur && c->fold_mode && c->fold_pending;   // 2D batch: the clock is part of this step's kernel (ClockFold)
  unsigned long long* slots = scan2 ? c->d_red : 0;
  if (st.clk && !(ND == 2 && scan2)) return -1;   // a device-clock step is a fused kernel with the CFL term or nothing
  if (scan2 && !c->clk_cur && rg_memset_async(c->d_red, 0, RG_DT_SLOTS * sizeof(unsigned long long), c->stream)) return -1;   // (the clock kernel zeroed them)
  if (ND == 2) {   // LDS-tiled fused step: one kernel (hip/tiled_hydro2d.h)
    Phase ph(c, RGPU_T_SWEEP);
    // plain faces, nothing modifying the new state after this kernel: it writes the ghost images too and the next step's fill is skipped
    int images = 0;
    if (rgpu::options().ghost_images && scan2 && !c->p.enableJet && g.nx >= g.gw && g.ny >= g.gw) {
      images = 1 << 12;
      for (int f = 0; f < 4; ++f) {
        const int bc = c->p.bc[f];
        if (bc != RGPU_BC_DIRICHLET && bc != RGPU_BC_NEUMANN && bc != RGPU_BC_PERIODIC) { images = 0; break; }
        images |= bc << (2 * f);
      }
    }
    const int rc = rgpu_tiled::hydro2d_step(c->stream, g, in, out, dtdx, dtdy, slots, images, folding ? 0 : st.clk, folding ? &c->fold : 0);
    if (rc == 0 && folding) c->fold_pending = false;
    if (rc == 0 && scan2) { c->fused_dt_parity = (out == c->U[0]) ? 0 : 1; c->fused_dt_slots = RG_DT_SLOTS; }
    if (rc == 0 && images) c->ghost_ok_parity = (out == c->U[0]) ? 0 : 1;
    if (rc <= 0) return rc;
    if (st.clk) return -1;   // the flat kernels take dt by value
  }
  { Phase ph(c, RGPU_T_PRIM); K_hydro_prim<NV> k = {g, in, c->Q}; if (launch_planes<kBlock, 1>(c->stream, g, clip(a - 2, b + 2, ks), k)) return -1; }
  const bool gf = g.grav_on == 2;   // per-cell gravity field: separate instantiations (see half_dt_gravity)
  int rc = 1;   // 1 = not handled by a specialisation
  if (rgpu::options().spec && g.grav_on == 0) {
    const int SL1 = SPEC_SLOPE1 | SPEC_NO_GRAVITY, SL2 = SPEC_SLOPE2 | SPEC_NO_GRAVITY;
    if (spec_matches(SPEC_HYDRO_APPROX | SL1, g)) rc = hydro_flux_trace_spec<ND, NV, SPEC_HYDRO_APPROX | SL1>(c, dtdx, dtdy, dtdz, a, b);
    else if (spec_matches(SPEC_HYDRO_APPROX | SL2, g)) rc = hydro_flux_trace_spec<ND, NV, SPEC_HYDRO_APPROX | SL2>(c, dtdx, dtdy, dtdz, a, b);
    else if (spec_matches(SPEC_HYDRO_HLLC | SL1, g)) rc = hydro_flux_trace_spec<ND, NV, SPEC_HYDRO_HLLC | SL1>(c, dtdx, dtdy, dtdz, a, b);
    else if (spec_matches(SPEC_HYDRO_HLLC | SL2, g)) rc = hydro_flux_trace_spec<ND, NV, SPEC_HYDRO_HLLC | SL2>(c, dtdx, dtdy, dtdz, a, b);
    else if (spec_matches(SPEC_HYDRO_HLL | SL1, g)) rc = hydro_flux_trace_spec<ND, NV, SPEC_HYDRO_HLL | SL1>(c, dtdx, dtdy, dtdz, a, b);
    else if (spec_matches(SPEC_HYDRO_HLL | SL2, g)) rc = hydro_flux_trace_spec<ND, NV, SPEC_HYDRO_HLL | SL2>(c, dtdx, dtdy, dtdz, a, b);
  }
  if (rc < 0) return -1;
  if (rc == 1) {
    { Phase ph(c, RGPU_T_TRACE); K_hydro_trace<ND, NV> k = {g, c->Q, c->T, dtdx, dtdy, dtdz}; if (launch_planes<kBlock, 1>(c->stream, g, clip(a - 1, b + 1, ks), k)) return -1; }
    Phase ph(c, RGPU_T_FLUX);
    K_hydro_flux<ND, NV, false> k = {g, c->T, c->F};
    K_hydro_flux<ND, NV, true> kg = {g, c->T, c->F};
    if (gf ? launch_planes<kBlockHeavy, 1>(c->stream, g, clip(a, b + 1, ks), kg) : launch_planes<kBlockHeavy, 1>(c->stream, g, clip(a, b + 1, ks), k)) return -1;
  }
  {
    Phase ph(c, RGPU_T_UPDATE);
    K_hydro_update<ND, NV, false> k = {g, in, out, c->F, dtdx, dtdy, dtdz, slots};
    K_hydro_update<ND, NV, true> kg = {g, in, out, c->F, dtdx, dtdy, dtdz, slots};
    if (gf ? launch_planes<kBlock, 1>(c->stream, g, clip(a, b, ks), kg) : launch_planes<kBlock, 1>(c->stream, g, clip(a, b, ks), k)) return -1;
  }
  if (scan2) { c->fused_dt_parity = (out == c->U[0]) ? 0 : 1; c->fused_dt_slots = RG_DT_SLOTS; }
  return 0;
}

// shearing-box / rotating-frame coefficients of the momentum update (MHDRunGodunov.cpp:2039-2053)
RotCoef rot_coef(const rgpu_ctx* c, double dt) {
  RotCoef rc = {0.0, 1.0, 1.0, 0.0};
  if (c->g.rot) {
    double lambda = c->p.Omega0 * dt;
    lambda = 0.25 * lambda * lambda;
    rc.lambda = lambda;
    rc.ratio = (1.0 - lambda) / (1.0 + lambda);
    rc.alpha1 = 1.0 / (1.0 + lambda);
    rc.alpha2 = c->p.Omega0 * dt / (1.0 + lambda);
  }
  return rc;
}

// Launch-time specialisations of the 3D MHD kernels (launchers.h): the isothermal rotating box (MRI) and the adiabatic
// inertial one, both with the HLLD pair, slope type 2 and no gravity; everything else runs the generic kernels.
const int kSpecMri = SPEC_HLLD | SPEC_ISOTHERMAL | SPEC_ROTATING | SPEC_NO_GRAVITY | SPEC_SLOPE2;
const int kSpecPlain = SPEC_HLLD | SPEC_ADIABATIC | SPEC_INERTIAL | SPEC_NO_GRAVITY | SPEC_SLOPE2;
inline int pick_spec(const DevParams& g) {
  return !rgpu::options().spec ? 0 : spec_matches(kSpecMri, g) ? 1 : spec_matches(kSpecPlain, g) ? 2 : 0;
}

int mhd2d_core(rgpu_ctx* c, const double* in, double* out, double dt_arg) {
  const DevParams& g = c->g;
  const StepTime st = step_time(c, dt_arg, 0.0);
  if (st.skip) return 0;
  const double dt = st.dt;
  const double dtdx = dt / g.dx, dtdy = dt / g.dy;
  const RotCoef rc = rot_coef(c, dt);
  {
    // LDS-tiled fused step (hip/tiled_mhd2d.h): U -> Unew in one kernel, the CFL term of the new state included under the
    // conditions of the flat update kernel below.  Not with a Dirichlet face (its ghost fill leaves B alone, so the output's
    // ghost cells must be copies of the input's: the flat update copies them, the fused kernel writes its own cells only).
    const rgpu_params& p = c->p;
    bool faces_ok = true;
    for (int f = 0; f < 4; ++f) faces_ok = faces_ok && (p.bc[f] == RGPU_BC_PERIODIC || p.bc[f] == RGPU_BC_NEUMANN);
    if (faces_ok && g.grav_on != 2) {
      bool scan = !(p.nu > 0) && !(p.eta > 0) && !p.randomForcingEnabled && !p.ouForcingEnabled;
      if (scan && g.rot) scan = p.bc[0] == RGPU_BC_PERIODIC && p.bc[1] == RGPU_BC_PERIODIC && p.bc[2] == RGPU_BC_PERIODIC && p.bc[3] == RGPU_BC_PERIODIC;
      if (rgpu_tiled::mhd2d_step_covers(g)) {
        if (st.clk && !scan) return -1;
        if (scan && !c->clk_cur && rg_memset_async(c->d_red, 0, RG_DT_SLOTS * sizeof(unsigned long long), c->stream)) return -1;   // (the clock kernel zeroed them)
        Phase ph(c, RGPU_T_SWEEP);
        // periodic box on the plain path, nothing modifying the new state after this kernel: it writes the periodic images too and
        // the next step's ghost fill is skipped (step_pre)
        bool images = rgpu::options().ghost_images && !g.rot && scan && !p.enableJet && g.nx >= g.gw && g.ny >= g.gw;
        for (int f = 0; f < 4; ++f) images = images && p.bc[f] == RGPU_BC_PERIODIC;
        const int rct = rgpu_tiled::mhd2d_step<kSpecPlain>(c->stream, g, rc, pick_spec(g) == 2, in, out, dt, scan ? c->d_red : 0, images ? 1 : 0, st.clk);
        if (rct < 0) return -1;
        if (rct == 0) {
          if (scan) { c->fused_dt_parity = (out == c->U[0]) ? 0 : 1; c->fused_dt_slots = RG_DT_SLOTS; }
          if (images) c->ghost_ok_parity = (out == c->U[0]) ? 0 : 1;
          return 0;
        }
      }
    }
  }
  if (st.clk) return -1;   // the flat kernels take dt by value
  { Phase ph(c, RGPU_T_PRIM); K_mhd_prim<> k = {g, in, c->Q, dt}; if (rg_launch<kBlock>(c->stream, c->n32, k)) return -1; }
  { Phase ph(c, RGPU_T_TRACE); K_mhd_trace2d k = {g, in, c->Q, c->T, dtdx, dtdy}; if (rg_launch<kBlock>(c->stream, c->n32, k)) return -1; }
  const bool gf = g.grav_on == 2;
  {
    Phase ph(c, RGPU_T_FLUX);
    K_mhd_flux2d<false> k = {g, c->T, c->F};
    K_mhd_flux2d<true> kg = {g, c->T, c->F};
    if (gf ? rg_launch<kBlockHeavy>(c->stream, c->n32, kg) : rg_launch<kBlockHeavy>(c->stream, c->n32, k)) return -1;
  }
  // the CFL scan of the new state rides in the update kernel under the conditions of the 3D step (mhd3d_core): nothing
  // modifies the state afterwards, and on the rotating path (ghosts refilled before the reference scans) the refilled high
  // faces are bit-identical periodic copies
  const rgpu_params& p = c->p;
  bool scan = !gf && !(p.nu > 0) && !(p.eta > 0) && !p.randomForcingEnabled && !p.ouForcingEnabled;
  if (scan && g.rot) scan = p.bc[0] == RGPU_BC_PERIODIC && p.bc[1] == RGPU_BC_PERIODIC && p.bc[2] == RGPU_BC_PERIODIC && p.bc[3] == RGPU_BC_PERIODIC;
  unsigned long long* slots = scan ? c->d_red : 0;
  if (scan && !c->clk_cur && rg_memset_async(c->d_red, 0, RG_DT_SLOTS * sizeof(unsigned long long), c->stream)) return -1;
  {
    Phase ph(c, RGPU_T_UPDATE);
    K_mhd_update2d<false> k = {g, rc, in, out, c->F, dt, dtdx, dtdy, slots};
    K_mhd_update2d<true> kg = {g, rc, in, out, c->F, dt, dtdx, dtdy, slots};
    if (gf ? rg_launch<kBlock>(c->stream, c->n32, kg) : rg_launch<kBlock>(c->stream, c->n32, k)) return -1;
  }
  if (scan) { c->fused_dt_parity = (out == c->U[0]) ? 0 : 1; c->fused_dt_slots = RG_DT_SLOTS; }
  return 0;
}

template <template <int> class K, int BLOCK, class... A>
int launch_spec(int spec, rg_stream_t s, const DevParams& g, PlaneRange r, A... a) {
  if (spec == 1) { K<kSpecMri> k = {g, a...}; return launch_planes<BLOCK, 1>(s, g, r, k); }
  if (spec == 2) { K<kSpecPlain> k = {g, a...}; return launch_planes<BLOCK, 1>(s, g, r, k); }
  K<SPEC_NONE> k = {g, a...};
  return launch_planes<BLOCK, 1>(s, g, r, k);
}
template <int S> using K_riemann_t = K_mhd_flux3d<DO_ALL, false, S>;

// 3D MHD: complete the update of planes [a,b).  The range is swept in chunks of ~8 planes; the HBM-bound stages
// (prim, elec, trace, update) go to the context stream, the fp64-VALU-bound Riemann stages (flux, emf) to a second
// stream, ordering-only events in between, so that trace of chunk c+1 runs next to flux/emf of chunk c.  With the
// phase timers on (or RGPU_CHUNKS=1) everything is issued on the context stream in one chunk.
// what: 0 = the whole update of planes [a,b); RGPU_CORE_FLUXES = only F, emf (+ the shear remap buffers) that update needs;
// RGPU_CORE_UPDATE = only the update, from F, emf computed by an earlier RGPU_CORE_FLUXES call covering [a,b)
// (a2, b2): a second plane range handled in the same call -- split calls only (the two boundary ranges of a slab): one launch of the
// update kernel for both
int mhd3d_core(rgpu_ctx* c, const double* in, double* out, double dt_arg, double t_arg, int a, int b, int what_flags = 0, int a2 = 0, int b2 = 0) {
  const StepTime st = step_time(c, dt_arg, t_arg);
  if (st.skip) return 0;
  const double dt = st.dt, totalTime = st.t;
  int what = what_flags;
  const DevParams& g = c->g;
  const rgpu_params& p = c->p;
  const double dtdx = dt / g.dx, dtdy = dt / g.dy, dtdz = dt / g.dz;
  const int ks = g.ksize;
  const RotCoef rc = rot_coef(c, dt);
  ShearRemap sr = {0, 0.0, 0.0};
  const bool shear = g.rot && g.shearbox;
  if (shear) {  // MHDRunGodunov.cpp:3213-3216 (flux / emf remap uses totalTime + dt/2)
    double deltay = 1.5 * p.Omega0 * (p.dx * p.nx) * (totalTime + dt / 2);
    deltay = std::fmod(deltay, (p.dy * p.ny));
    sr.jplus = (int)(deltay / p.dy);
    const double epsi = std::fmod(deltay, p.dy);
    sr.eps_min = 1.0 - epsi / p.dy;
    sr.eps_max = epsi / p.dy;
  }
  // (one launch for the three face (HLLD) and the three edge (2D HLLD) Riemann problems of a cell: they read the same
  // traced states T, and T is 60 % of the step's HBM traffic.  Two launches -- 128 VGPRs / 4 waves per SIMD for the
  // faces, 205 / 2 for the edges -- were faster while the solvers were purely VALU bound; after the shared-reciprocal
  // rewrite and the XCD-aware order the second read of T costs more: 64.4 -> 60.9 ms/step at 512^3.)
  const bool gf = g.grav_on == 2;   // per-cell gravity field: its own instantiations (see half_dt_gravity)
  const int spec = gf ? 0 : pick_spec(g);
  const double* Q = c->Q; const double* E = c->E; const double* T = c->T; const double* F = c->F; const double* emf = c->emf;
  const double* remap = c->shear_remap;
  auto prim_planes = [&](rg_stream_t s, PlaneRange r) -> int { return launch_spec<K_mhd_prim, kBlock>(spec, s, g, r, in, c->Q, dt); };
  auto elec_planes = [&](rg_stream_t s, PlaneRange r) -> int { return launch_spec<K_mhd_elec, kBlock>(spec, s, g, r, in, Q, c->E); };
  auto trace_planes = [&](rg_stream_t s, PlaneRange r) -> int { return launch_spec<K_mhd_trace3d, kBlock>(spec, s, g, r, in, Q, E, c->T, dtdx, dtdy, dtdz); };
  auto riemann_planes = [&](rg_stream_t s, PlaneRange r) -> int {
    if (gf) { K_mhd_flux3d<DO_ALL, true> k = {g, T, c->F, c->emf}; return launch_planes<kBlockHeavy, 1>(s, g, r, k); }
    return launch_spec<K_riemann_t, kBlockHeavy>(spec, s, g, r, T, c->F, c->emf);
  };
  // LDS-tiled fused trace + Riemann sweep over the Riemann planes r (hip/tiled_mhd.h); 1 = not covered
  // r2: a second range in the same launch (split calls on the two boundary ranges of a slab) -- taken when both ranges clip to the
  // same number of planes; returns 2 when it was not (the caller launches range by range)
  auto sweep_planes = [&](rg_stream_t s, PlaneRange r, PlaneRange r2 = PlaneRange{0, 0}) -> int {
    const int lo = r.lo < g.gw ? g.gw : r.lo, hi = r.hi > ks - g.gw + 1 ? ks - g.gw + 1 : r.hi;
    int lo2 = 0;
    if (r2.hi > r2.lo) {
      lo2 = r2.lo < g.gw ? g.gw : r2.lo;
      const int hi2 = r2.hi > ks - g.gw + 1 ? ks - g.gw + 1 : r2.hi;
      if (hi <= lo || hi2 - lo2 != hi - lo || lo2 < hi) return 2;
    }
    // periodic faces whose fluxes / EMFs are bit-identical copies of the opposite layer (see K_copy_periodic_layer): y when both
    // y faces are periodic; x when both x faces are periodic and the frame does not rotate (the rotating-frame terms carry xPos)
    int reuse = 0;
    if (p.bc[2] == RGPU_BC_PERIODIC && p.bc[3] == RGPU_BC_PERIODIC) reuse |= 2;
    if (p.bc[0] == RGPU_BC_PERIODIC && p.bc[1] == RGPU_BC_PERIODIC && !g.rot) reuse |= 1;
    // shearing box: the launch that copies the periodic y layer also saves the emfY border columns of these planes for the remap
    return rgpu_tiled::mhd3d_sweep<kSpecMri, kSpecPlain>(s, g, spec, in, c->F, c->emf, dt, dtdx, dtdy, dtdz, lo, hi, reuse, st.clk, shear ? c->shear_save : 0, lo2);
  };
  // trace of planes [t_lo, t_hi) + Riemann problems of planes rf: fused when the backend covers the configuration
  const bool use_sweep = !gf && rgpu_tiled::mhd3d_sweep_covers(g);
  if (st.clk && !use_sweep) return -1;   // the flat prim / elec / trace / Riemann kernels take dt by value
  auto trace_riemann = [&](rg_stream_t s, int t_lo, int t_hi, PlaneRange rf) -> int {
    if (use_sweep) { Phase ph(c, RGPU_T_SWEEP); return sweep_planes(s, rf); }
    { Phase ph(c, RGPU_T_TRACE); if (trace_planes(s, clip(t_lo, t_hi, ks))) return -1; }
    { Phase ph(c, RGPU_T_FLUX); if (riemann_planes(s, rf)) return -1; }
    return 0;
  };
  K_shear_save_emf k_ssave = {g, c->emf, c->shear_save};
  K_shear_remap k_sremap = {g, sr, c->F, c->emf, c->shear_save, c->shear_remap, dtdx, st.clk};
  auto shear_planes = [&](rg_stream_t s, PlaneRange r, PlaneRange r2 = PlaneRange{0, 0}) -> int {  // the two 2D (j,k) kernels restricted to planes r (and r2)
    if (!shear || r.hi <= r.lo) return 0;
    if (use_sweep) {   // the sweep's closing launch saved the emfY columns of its planes: the remap of exactly those
      if (r.lo < g.gw) r.lo = g.gw;
      if (r.hi > ks - g.gw + 1) r.hi = ks - g.gw + 1;
      if (r.hi <= r.lo) return 0;
      if (r2.hi > r2.lo) {   // both boundary ranges in one launch (after a two-range sweep)
        if (r2.lo < g.gw) r2.lo = g.gw;
        if (r2.hi > ks - g.gw + 1) r2.hi = ks - g.gw + 1;
        const unsigned n1 = (unsigned)(r.hi - r.lo) * g.jsize, n2 = (unsigned)(r2.hi - r2.lo) * g.jsize;
        K_two_ranges<K_shear_remap> k2 = {k_sremap, (unsigned)r.lo * g.jsize, n1, (unsigned)r2.lo * g.jsize};
        return rg_launch<kBlock>(s, n1 + n2, k2);
      }
      return rg_launch_range<kBlock>(s, (unsigned)r.lo * g.jsize, (unsigned)(r.hi - r.lo) * g.jsize, k_sremap);
    }
    const unsigned j0 = (unsigned)r.lo * g.jsize, jn = (unsigned)(r.hi - r.lo) * g.jsize;
    return rg_launch_range<kBlock>(s, j0, jn, k_ssave) || rg_launch_range<kBlock>(s, j0, jn, k_sremap);
  };
  // The CFL scan of the new state rides in the update kernel when the whole domain is updated in one call and the
  // next compute_dt will see exactly this state: nothing modifies it afterwards (no dissipative stage / forcing), and the
  // field on the three high boundary faces keeps its CT value -- always true on the plain path (the reference scans
  // before the ghosts are refilled), on the rotating path when y, z are periodic (the refilled faces are bit-identical
  // copies) and x is periodic or the shearing box (its ghost fill skips the first outer Bx face).
  // Slab pieces (RGPU_CORE_SCAN with the split calls): the same scan accumulated over the update launches of a step -- the
  // slots are reset by the FLUXES call; a z face shared with a neighbour slab (RGPU_BC_COPY) counts like a periodic one: the
  // exchanged faces are the doubles this slab's own CT update gives them.
  const bool acc = (what & RGPU_CORE_SCAN) != 0;
  what &= ~RGPU_CORE_SCAN;
  const bool cond = mhd3d_scan_cond(c);
  const int out_parity = (out == c->U[0]) ? 0 : 1;
  bool scan = what == 0 && a <= 0 && b >= ks && cond;
  if (scan && g.rot && (p.bc[4] == RGPU_BC_COPY || p.bc[5] == RGPU_BC_COPY)) scan = false;   // whole-slab call of a slab: the driver scans
  if (acc && what == RGPU_CORE_FLUXES) {
    c->scan_acc_parity = cond ? out_parity : -1;
    if (cond && !c->clk_cur && rg_memset_async(c->d_red, 0, RG_DT_SLOTS * sizeof(unsigned long long), c->stream)) return -1;   // (a clock kernel zeroed them)
  }
  const bool scan_piece = acc && what == RGPU_CORE_UPDATE && cond && c->scan_acc_parity == out_parity;
  unsigned long long* slots = (scan || scan_piece) ? c->d_red : 0;
  if (scan && !c->clk_cur && rg_memset_async(c->d_red, 0, RG_DT_SLOTS * sizeof(unsigned long long), c->stream)) return -1;
  // the update is a pure stream over F, emf and U: one thread per column and short z segment, linear workgroup order, the plane
  // k+1 entries carried in registers (mhd_update3d_column; 512^3: 8.07 -> 7.42 ms against one thread per cell)
  const int upd_seg = 3;   // planes per thread of the update's z march (512^3: 2 / 3 / 4 / 8 / 32 planes 7.49 / 7.42 / 7.50 / 7.65 / 9.0 ms)
  auto update_planes = [&](rg_stream_t s, PlaneRange r, PlaneRange r2 = PlaneRange{0, 0}) -> int {
    if (r.hi <= r.lo) { r = r2; r2 = PlaneRange{0, 0}; }
    if (r.hi <= r.lo) return 0;
    const int seg_len = upd_seg;
    const unsigned n1 = g.sk * (unsigned)((r.hi - r.lo + seg_len - 1) / seg_len);
    const bool two = r2.hi > r2.lo;
    const unsigned nt = n1 + (two ? g.sk * (unsigned)((r2.hi - r2.lo + seg_len - 1) / seg_len) : 0u);
    const unsigned split = two ? n1 : 0xffffffffu;
#define RG_UPD(ROT, GF, S) { K_mhd_update3d<ROT, GF, S> k = {g, rc, in, out, F, emf, remap, dt, dtdx, dtdy, dtdz, slots, r.lo, r.hi, seg_len, split, r2.lo, r2.hi, st.clk}; return rg_launch_range<kBlock>(s, 0u, nt, k); }
    if (gf) { if (g.rot) RG_UPD(true, true, SPEC_NONE) else RG_UPD(false, true, SPEC_NONE) }
    if (g.rot) { if (spec == 1) RG_UPD(true, false, kSpecMri) if (spec == 2) RG_UPD(true, false, kSpecPlain) RG_UPD(true, false, SPEC_NONE) }
    if (spec == 1) RG_UPD(false, false, kSpecMri) if (spec == 2) RG_UPD(false, false, kSpecPlain) RG_UPD(false, false, SPEC_NONE)
#undef RG_UPD
  };

  // the fused sweep marches along z inside one launch: cutting the range into chunks only adds prologues (measured 60.4
  // against 55.3 ms/step at 512^3), so the two-stream chunk schedule is kept for the flat kernels only
  const bool serial = what != 0 || c->timers_on || c->nchunks <= 1 || (b - a) < 16 || use_sweep || c->clk_cur;
  const bool pair = what != 0 && b2 > a2;
  if (serial) {
    rg_stream_t s = c->stream;
    bool fluxes_done = false;
    if (what != RGPU_CORE_UPDATE && pair && use_sweep) {   // both boundary ranges: one launch of the sweep, one of the remap
      Phase ph(c, RGPU_T_SWEEP);
      const int rcs = sweep_planes(s, clip(a, b + 1, ks), clip(a2, b2 + 1, ks));
      if (rcs < 0 || rcs == 1) return -1;
      fluxes_done = rcs == 0;
    }
    if (fluxes_done) { Phase ph(c, RGPU_T_SHEAR); if (shear_planes(s, clip(a, b + 1, ks), clip(a2, b2 + 1, ks))) return -1; }
    if (what != RGPU_CORE_UPDATE && !fluxes_done) {
      for (int n = 0; n < (pair ? 2 : 1); ++n) {
        const int lo = n ? a2 : a, hi = n ? b2 : b;
        if (!use_sweep) {   // the sweep computes primitives and electric field itself (in LDS)
          { Phase ph(c, RGPU_T_PRIM); if (prim_planes(s, clip(lo - 2, hi + 2, ks))) return -1; }
          { Phase ph(c, RGPU_T_ELEC); if (elec_planes(s, clip(lo - 1, hi + 2, ks))) return -1; }
        }
        if (trace_riemann(s, lo - 1, hi + 1, clip(lo, hi + 1, ks))) return -1;
        { Phase ph(c, RGPU_T_SHEAR); if (shear_planes(s, clip(lo, hi + 1, ks))) return -1; }
      }
    }
    if (what != RGPU_CORE_FLUXES) {
      { Phase ph(c, RGPU_T_UPDATE); if (update_planes(s, clip(a, b, ks), pair ? clip(a2, b2, ks) : PlaneRange{0, 0})) return -1; }
      if (scan) { c->fused_dt_parity = (out == c->U[0]) ? 0 : 1; c->fused_dt_slots = RG_DT_SLOTS; }
    }
    return 0;
  }

  // chunked two-stream schedule: stage s has completed planes [.., done_s); per chunk each stage advances to what
  // the update of planes < kb needs
  rg_stream_t sm = c->stream, sa = c->stream2;
  const int span = b - a;
  int C = (span + 7) / 8;
  if (C > c->nchunks) C = c->nchunks;
  if (C < 1) C = 1;
  if (rg_event_record(c->ev_fork, sm) || rg_stream_wait_event(sa, c->ev_fork)) return -1;
  int d_prim = a - 2, d_elec = a - 1, d_trace = a - 1, d_flux = a, d_upd = a;
  for (int ci = 0; ci <= C; ++ci) {
    if (ci < C) {
      const int kb = (ci + 1 == C) ? b : a + (int)(((long long)span * (ci + 1)) / C);
      if (!use_sweep && prim_planes(sm, clip(d_prim, kb + 2, ks))) return -1;
      d_prim = kb + 2;
      if (!use_sweep && elec_planes(sm, clip(d_elec, kb + 2, ks))) return -1;
      d_elec = kb + 2;
      if (!use_sweep && trace_planes(sm, clip(d_trace, kb + 1, ks))) return -1;
      d_trace = kb + 1;
      if (rg_event_record(c->ev_trace[ci], sm) || rg_stream_wait_event(sa, c->ev_trace[ci])) return -1;
      const PlaneRange rf = clip(d_flux, kb + 1, ks);
      if (use_sweep ? sweep_planes(sa, rf) : riemann_planes(sa, rf)) return -1;
      if (shear_planes(sa, rf)) return -1;
      d_flux = kb + 1;
      if (rg_event_record(c->ev_flux[ci], sa)) return -1;
    }
    if (ci >= 1) {  // update lags one chunk so that the next chunk's prim/elec/trace are queued ahead of it
      const int kb_prev = (ci == C) ? b : a + (int)(((long long)span * ci) / C);
      if (rg_stream_wait_event(sm, c->ev_flux[ci - 1])) return -1;
      if (update_planes(sm, clip(d_upd, kb_prev, ks))) return -1;
      d_upd = kb_prev;
    }
  }
  if (scan) { c->fused_dt_parity = (out == c->U[0]) ? 0 : 1; c->fused_dt_slots = RG_DT_SLOTS; }
  return 0;
}

// what != 0 (RGPU_CORE_FLUXES / RGPU_CORE_UPDATE) splits the 3D MHD step, the only one whose update is a kernel of its own;
// for every other solver FLUXES is a no-op and UPDATE the whole piece, so a driver may use the split schedule blindly
int step_core_planes(rgpu_ctx* c, int nStep, double dt, double totalTime, int a, int b, int what = 0, int a2 = 0, int b2 = 0) {
  const bool splittable = c->g.three_d && c->p.mhdEnabled;
  const bool acc = (what & RGPU_CORE_SCAN) != 0;
  bool hydro_piece = false;
  if ((what & ~RGPU_CORE_SCAN) != 0 && !splittable) {
    if ((what & ~RGPU_CORE_SCAN) == RGPU_CORE_FLUXES) {   // nothing to compute; with SCAN: reset the slot for the pieces that follow
      c->scan_acc_parity = -1;
      if (acc && c->g.three_d && !c->p.mhdEnabled && !(c->p.nu > 0) && !c->p.randomForcingEnabled && !c->p.ouForcingEnabled &&
          rgpu_tiled::hydro3d_sweep_covers(c->g) && c->p.gravityEnabled != 2) {
        if (!c->clk_cur && rg_memset_async(c->d_red, 0, RG_DT_SLOTS * sizeof(unsigned long long), c->stream)) return -1;   // (a clock kernel zeroed them)
        c->scan_acc_parity = (nStep + 1) % 2;
      }
      return 0;
    }
    hydro_piece = acc && c->g.three_d && !c->p.mhdEnabled;
    what = 0;
  }
  c->fused_dt_parity = -1;   // the output array is about to change (a whole-domain hydro sweep sets it again)
  c->ghost_ok_parity = -1;
  // static gravity of this step: (0.5 * dt) * g, the reference's "HALF_F * dt * h_gravity"; of the 2D MHD steps only
  // implementation version 0 has it
  c->g.grav_on = (c->p.gravityEnabled && !(c->p.mhdEnabled && !c->g.three_d && (c->p.implementationVersion != 0 || c->g.rot))) ? 1 : 0;
  if (c->g.grav_on && c->p.gravityEnabled == 2) c->g.grav_on = 2;   // per-cell field (rgpu_set_gravity_field)
  c->g.G = c->G;
  c->g.hdt = 0.5 * dt;
  c->g.hgx = 0.5 * dt * c->p.gravity_x;
  c->g.hgy = 0.5 * dt * c->p.gravity_y;
  c->g.hgz = 0.5 * dt * c->p.gravity_z;
  const double* in = c->U[nStep % 2];
  double* out = c->U[(nStep + 1) % 2];
  if (!c->g.three_d) {  // 2D: no planes
    if (!c->p.mhdEnabled) return hydro_core<2, 4>(c, in, out, dt, 0, 1);
    return mhd2d_core(c, in, out, dt);
  }
  if (a < 0) a = 0;
  if (b > c->g.ksize) b = c->g.ksize;
  if (a2 < 0) a2 = 0;
  if (b2 > c->g.ksize) b2 = c->g.ksize;
  if (b <= a) { a = a2; b = b2; a2 = b2 = 0; }
  if (b <= a) return 0;
  if (!c->p.mhdEnabled) {   // the sweep is the whole step: both ranges in one launch of the tiled sweep, else range by range
    if (b2 > a2 && rgpu_tiled::hydro3d_sweep_covers(c->g) && c->g.grav_on != 2) return hydro_core<3, 5>(c, in, out, dt, a, b, hydro_piece, a2, b2);
    const int rc = hydro_core<3, 5>(c, in, out, dt, a, b, hydro_piece);
    if (rc || b2 <= a2) return rc;
    return hydro_core<3, 5>(c, in, out, dt, a2, b2, hydro_piece);
  }
  if (what == 0 && b2 > a2) return mhd3d_core(c, in, out, dt, totalTime, a, b, 0) || mhd3d_core(c, in, out, dt, totalTime, a2, b2, 0);
  return mhd3d_core(c, in, out, dt, totalTime, a, b, what, a2, b2);
}

// Dissipative stage ([hydro] nu, [MHD] eta) on the state the step has just written: refill its ghosts (plain or
// shearing-box fill, as the call sites do), resistive emf + CT (+ energy flux unless isothermal), then viscous fluxes.
// Scratch: fluxes in F, the resistive emf in T (both dead at this point of the step).
template <int ND>
int dissipative_nd(rgpu_ctx* c, double* U, double dt, double nu, double eta) {
  const DevParams& g = c->g;
  const unsigned n = c->n32;
  if (eta > 0) {
    K_resist_emf<ND> ke = {g, U, c->T, eta};
    K_resist_ct<ND> kc = {g, U, c->T, dt / g.dx, dt / g.dy, dt / g.dz};
    if (rg_launch<kBlock>(c->stream, n, ke) || rg_launch<kBlock>(c->stream, n, kc)) return -1;
    if (g.cIso <= 0) {
      K_resist_eflux<ND> kf = {g, U, c->F, eta, dt};
      K_flux_update<ND> ku = {g, U, c->F, IP, IP + 1};
      if (rg_launch<kBlock>(c->stream, n, kf) || rg_launch<kBlock>(c->stream, n, ku)) return -1;
    }
  }
  if (nu > 0) {
    K_visc_flux<ND> kv = {g, U, c->F, nu, dt};
    K_flux_update<ND> ku = {g, U, c->F, 0, ND + 2};
    if (rg_launch<kBlock>(c->stream, n, kv) || rg_launch<kBlock>(c->stream, n, ku)) return -1;
  }
  return 0;
}

// Every entry point that WRITES a state array outside the step kernels calls this: what the context remembers about that state --
// the CFL maximum a kernel left in the device slots (fused_dt_parity), a scan being accumulated piece by piece
// (scan_acc_parity), ghost cells the step kernel wrote itself (ghost_ok_parity) -- is void from here on.
inline void state_modified(rgpu_ctx* c) {
  c->fused_dt_parity = -1;
  c->scan_acc_parity = -1;
  c->ghost_ok_parity = -1;
}

int step_dissipative(rgpu_ctx* c, int nStep, double dt, double totalTime, bool fill_ghosts = true) {
  const double nu = c->p.nu, eta = c->p.mhdEnabled ? c->p.eta : 0.0;
  if (!(nu > 0 || eta > 0)) return 0;
  state_modified(c);
  Phase ph(c, RGPU_T_DISSIPATIVE);
  double* U = c->U[(nStep + 1) % 2];
  int rc = 0;
  if (!fill_ghosts) {
    // slab driver: it has filled the ghosts itself (in-plane fills + z exchange)
  } else if (c->g.shearbox && c->g.three_d) {
    rc = do_make_boundaries(c, U, RGPU_YDIR) || do_make_boundaries_shear(c, U, totalTime, dt) ||
         do_make_boundaries(c, U, RGPU_ZDIR) || do_make_boundaries(c, U, RGPU_YDIR);
  } else {
    rc = do_make_boundaries(c, U, RGPU_XDIR) || do_make_boundaries(c, U, RGPU_YDIR) || (c->g.three_d && do_make_boundaries(c, U, RGPU_ZDIR));
  }
  if (rc) return -1;
  return c->g.three_d ? dissipative_nd<3>(c, U, dt, nu, eta) : dissipative_nd<2>(c, U, dt, nu, eta);
}

int step_core(rgpu_ctx* c, int nStep, double dt, double totalTime) {
  return step_core_planes(c, nStep, dt, totalTime, 0, c->g.ksize);
}

// max of the per-cell 1/dt over the flat index range [idx0, idx0+n) into the device slot (reset or accumulate)
int inv_dt_scan(rgpu_ctx* c, int parity, unsigned idx0, unsigned n, bool reset) {
  c->fused_dt_parity = -1;   // the slot is rewritten
  c->scan_acc_parity = -1;
  Phase ph(c, RGPU_T_DT);
  const double* U = c->U[parity & 1];
  // a fresh scan owns ALL slots: the maximum goes to slot 0, slots 1 .. RG_DT_SLOTS-1 (which a fused scan of an earlier step may
  // have filled) are zeroed, so that whoever folds all of them -- the slab driver after its fixed-size all-reduce, whatever state
  // each rank is in -- reads this scan and nothing older
  if (reset && rg_memset_async(c->d_red, 0, RG_DT_SLOTS * sizeof(unsigned long long), c->stream)) return -1;
  reset = false;
  if (c->p.mhdEnabled) {
    const int spec = c->g.three_d ? pick_spec(c->g) : 0;
    if (spec == 1) { K_mhd_invdt<kSpecMri> k = {c->g, U}; return rg_reduce_max(c->stream, n, k, c->d_red, idx0, reset); }
    if (spec == 2) { K_mhd_invdt<kSpecPlain> k = {c->g, U}; return rg_reduce_max(c->stream, n, k, c->d_red, idx0, reset); }
    K_mhd_invdt<> k = {c->g, U};
    return rg_reduce_max(c->stream, n, k, c->d_red, idx0, reset);
  }
  if (c->g.three_d) { K_hydro_invdt<5> k = {c->g, U}; return rg_reduce_max(c->stream, n, k, c->d_red, idx0, reset); }
  K_hydro_invdt<4> k = {c->g, U};
  return rg_reduce_max(c->stream, n, k, c->d_red, idx0, reset);
}

int inv_dt_fetch(rgpu_ctx* c, double* invDt, int nslots = 1) {
  if (rg_copy_d2h(c->h_red, c->d_red, (size_t)nslots * sizeof(unsigned long long), c->stream) || rg_stream_sync(c->stream)) return -1;
  double v = 0.0;
  for (int s = 0; s < nslots; ++s) {
    double x;
    std::memcpy(&x, c->h_red + s, sizeof(double));
    v = std::fmax(v, x);
  }
  // seeds and jet term of the CPU paths (HydroRunBase.cpp:382,420-422 ; MHDRunBase.cpp:144,184-186,228-231)
  const rgpu_params& p = c->p;
  if (p.mhdEnabled) v = std::fmax(v, p.smallc / std::fmin(p.dx, p.dy));
  if (p.enableJet) v = std::fmax(v, (p.ujet + p.cjet) / p.dx);
  *invDt = v;
  return 0;
}

int inv_dt(rgpu_ctx* c, int parity, double* invDt) {
  if (c->fused_dt_parity == (parity & 1)) return inv_dt_fetch(c, invDt, c->fused_dt_slots);   // the kernel that wrote this state scanned it
  return inv_dt_scan(c, parity, 0, c->n32, true) || inv_dt_fetch(c, invDt);
}

}  // namespace
