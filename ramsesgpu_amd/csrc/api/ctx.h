// api/ctx.h -- the context of the C ABI (struct rgpu_ctx), parameter validation, allocation, creation.  Part of the ONE translation
// unit rgpu_api.cpp (included there, in this order: ctx, boundaries, step, history, then the entry points).
#pragma once
namespace {
const int kBlock = 256;      // streaming kernels
const int kBlockHeavy = 64;  // Riemann kernels: 256 VGPRs, one wave per workgroup places best (64: 61.8, 128: 62.6, 256: 71.4 ms/step)
}

struct rgpu_ctx {
  rgpu_params p;
  DevParams g;
  rg_stream_t stream;
  bool own_state;
  double* U[2];
  double *Q, *E, *T, *F, *emf, *shear_save, *shear_remap;
  double* G;   // per-cell static gravity field (gravityEnabled == 2), 3 components
  double* Frc; // static driving field of the "turbulence" problem (randomForcingEnabled), 3 components
  rgpu_ou::OuProcess* ou;   // Ornstein-Uhlenbeck forcing process (ouForcingEnabled)
  unsigned long long* d_red;
  unsigned long long* h_red;
  size_t ncell, scratch_bytes;
  unsigned n32;
  // instrumentation
  bool timers_on;
  double t_acc[RGPU_T_COUNT];
  long t_calls[RGPU_T_COUNT];
  rg_event_t ev0, ev1;
  bool ev_ok;
  // z-chunked two-stream schedule of the 3D MHD step (mhd3d_core_overlap)
  enum { kMaxChunks = 256 };
  int nchunks;
  rg_stream_t stream2;
  rg_event_t ev_fork, ev_trace[kMaxChunks], ev_flux[kMaxChunks];
  int n_order_events;   // ev_trace / ev_flux pairs actually created (freed in rgpu_destroy whatever nchunks became)
  bool fork_ok;
  int device;           // HIP device the context was created on; every entry point makes it current
  unsigned xcd_sub;     // sub-band size (cells) of the XCD-aware workgroup order of THIS context, 0 = linear
  int fused_dt_parity;  // parity of the state whose CFL maximum the last sweep left in d_red (-1: none)
  int fused_dt_slots;   // how many slots of d_red hold it (1: hydro sweep; RG_DT_SLOTS: MHD update kernel)
  int ghost_ok_parity;  // parity of the state whose ghost cells the step kernel itself left valid (2D MHD, periodic box: images written
                        // by the fused kernel), -1: none -- the plain path then skips the ghost fill of that state at the next step's entry
  int scan_acc_parity;  // parity of the state whose CFL maximum is being accumulated piece by piece (RGPU_CORE_SCAN), -1: none
  // device-side time step (hip/step_clock.h; rgpu_run_steps): records of a batch on the device / pinned host memory, and the record the
  // step being queued reads (0: the step takes its by-value dt arguments)
  enum { kClockBatch = RGPU_CLOCK_BATCH };
  StepClock* d_clk; StepClock* h_clk; const StepClock* clk_cur;
  int clk_n;                    // records queued in the open batch (rgpu_clock_open .. rgpu_clock_close), -1: no batch open
  double clk_t0, clk_tEnd;
  // fused 2D steps: the clock is folded into the step kernel itself (step_clock_rec.h: ClockFold) over three rotating slot arrays;
  // d_red always points at the array that holds the maxima of the current state
  unsigned long long* d_red_base;   // 3 x RG_DT_SLOTS
  bool fold_mode, fold_pending; int fold_phase0; ClockFold fold;
  bool fold_request;            // set by rgpu_run_steps_log around its own rgpu_clock_open (clock_ready holds there): a batch opened from outside never folds
  std::string err;
};

namespace {

int fail(rgpu_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  return code;
}

// ---- the time step of the step being queued ------------------------------------------------------------------------
// By value from the caller -- or, inside a batch of device-clock steps (rgpu_clock_open .. close; csrc/step_clock_rec.h), the record
// c->clk_cur: the kernels that depend on dt read it on the device (st.clk), the host's copies are unused.  The test-only host
// emulation runs every "launch" at once, so there the record is already filled in and is resolved here, by value, for all kernels.
struct StepTime { double dt, t; const StepClock* clk; bool skip; };
inline StepTime step_time(const rgpu_ctx* c, double dt, double t) {
  StepTime st = {dt, t, 0, false};
  if (!c->clk_cur) return st;
#if RG_SYNC_LAUNCH
  st.dt = c->clk_cur->dt; st.t = c->clk_cur->t_cur; st.skip = c->clk_cur->stop != 0;
#else
  st.dt = 0.0; st.t = 0.0; st.clk = c->clk_cur;
#endif
  return st;
}
// the record for kernels that take nothing from it but "this step does not run"
inline const StepClock* stop_clk(const rgpu_ctx* c) { return RG_SYNC_LAUNCH ? 0 : c->clk_cur; }
inline bool stop_now(const rgpu_ctx* c) { return RG_SYNC_LAUNCH && c->clk_cur && c->clk_cur->stop != 0; }

// ---- phase timer: events around one phase; resolved immediately (timers serialise the stream by design) ----
struct Phase {
  rgpu_ctx* c; int which;
  Phase(rgpu_ctx* ctx, int w) : c(ctx), which(w) { if (c->timers_on && c->ev_ok) rg_event_record(c->ev0, c->stream); }
  ~Phase() {
    if (c->timers_on && c->ev_ok) {
      rg_event_record(c->ev1, c->stream);
      c->t_acc[which] += rg_event_elapsed_ms(c->ev0, c->ev1) * 1e-3;
      c->t_calls[which] += 1;
    }
  }
};

int validate(const rgpu_params* p, std::string* why) {
  if (!p) { *why = "params is NULL"; return RGPU_EINVAL; }
  if (p->abi_version != RGPU_ABI_VERSION) { *why = "abi_version mismatch"; return RGPU_EINVAL; }
  const bool three_d = p->nz_global != 1;
  const int gw_needed = p->mhdEnabled ? 3 : 2;
  if (p->ghostWidth < gw_needed || p->ghostWidth > 3) { *why = "ghostWidth must be 2 (hydro) or 3 (MHD)"; return RGPU_EINVAL; }
  if (p->nx < p->ghostWidth || p->ny < p->ghostWidth || (three_d && p->nz < p->ghostWidth)) { *why = "domain thinner than the ghost width"; return RGPU_EINVAL; }
  const int nv = p->mhdEnabled ? 8 : (three_d ? 5 : 4);
  if (p->nbVar != nv) { *why = "nbVar inconsistent with MHD / dimension"; return RGPU_EINVAL; }
  if (!(p->slope_type == 0 || p->slope_type == 1 || p->slope_type == 2 || p->slope_type == 3)) { *why = "slope_type must be 0, 1, 2 or 3"; return RGPU_EINVAL; }
  // positivity preserving slopes exist in the 2D MHD and the plain 3D MHD steps only: the hydro steps and the rotating
  // 3D step call slope routines that leave dq unset for type 3 (slope.h:97-147,324-427; slope_mhd.h:436-502)
  if (p->slope_type == 3 && (!p->mhdEnabled || (p->Omega0 > 0 && p->nz_global != 1))) { *why = "slope_type 3 is defined for 2D MHD and non-rotating 3D MHD only (the reference leaves the slopes unset elsewhere)"; return RGPU_EUNSUPPORTED; }
  if (p->mhdEnabled) {
    // 2D: versions 0 and 1 compute the same numbers (0 recomputes what 1 stores; 0 alone has the gravity terms); 2 is a
    // superseded variant
    if (!three_d && p->implementationVersion != 1 && p->implementationVersion != 0) { *why = "2D MHD: implementationVersion must be 0 or 1"; return RGPU_EUNSUPPORTED; }
    if (three_d && !(p->Omega0 > 0) && p->implementationVersion != 3 && p->implementationVersion != 4) { *why = "3D MHD: only implementationVersion 3/4 are implemented"; return RGPU_EUNSUPPORTED; }
    if (p->magRiemannSolver != RGPU_MAG_HLLD && p->magRiemannSolver != RGPU_MAG_HLLF && p->magRiemannSolver != RGPU_MAG_HLLA &&
        p->magRiemannSolver != RGPU_MAG_LLF) { *why = "magRiemannSolver must be hlld, hllf, hlla or llf (roe / upwind do not exist in the reference either)"; return RGPU_EUNSUPPORTED; }
    if (p->shearingBoxEnabled && !three_d) { *why = "shearing box needs 3D"; return RGPU_EUNSUPPORTED; }
  } else {
    if (p->unsplitVersion != 1 && p->unsplitVersion != 2) { *why = "hydro: unsplitVersion must be 1 or 2 (version 0 is a superseded variant)"; return RGPU_EUNSUPPORTED; }
    if (p->riemannSolver != RGPU_RS_APPROX && p->riemannSolver != RGPU_RS_HLL && p->riemannSolver != RGPU_RS_HLLC) { *why = "hydro riemannSolver must be approx, hll or hllc"; return RGPU_EINVAL; }
  }
  if (p->nu < 0 || p->eta < 0) { *why = "nu and eta must be >= 0"; return RGPU_EINVAL; }
  if (p->gravityEnabled < 0 || p->gravityEnabled > 2) { *why = "gravityEnabled must be 0, 1 (uniform vector) or 2 (per-cell field)"; return RGPU_EINVAL; }
  if (p->randomForcingEnabled && (!three_d || (p->mhdEnabled && p->Omega0 > 0))) { *why = "random forcing exists in the 3D non-rotating steps only (as in the reference)"; return RGPU_EUNSUPPORTED; }
  if (p->ouForcingEnabled && (!three_d || (p->mhdEnabled && p->Omega0 > 0))) { *why = "Ornstein-Uhlenbeck forcing exists in the 3D non-rotating steps only (as in the reference)"; return RGPU_EUNSUPPORTED; }
  if (p->ouForcingEnabled && !(p->ouTimeScaleTurb > 0)) { *why = "ouTimeScaleTurb must be > 0"; return RGPU_EINVAL; }
  for (int f = 0; f < 6; ++f) {
    const int b = p->bc[f];
    const bool ok = b == RGPU_BC_DIRICHLET || b == RGPU_BC_NEUMANN || b == RGPU_BC_PERIODIC || b == RGPU_BC_COPY ||
                    (b == RGPU_BC_SHEARINGBOX && f < 2) ||
                    (b == RGPU_BC_Z_STRATIFIED && f >= 4 && three_d && p->mhdEnabled && p->ghostWidth == 3 && p->cIso > 0 && p->Omega0 > 0);
    if (!ok && (three_d || f < 4)) { *why = "unsupported boundary condition type (z-stratified: z faces of an isothermal rotating 3D MHD box only)"; return RGPU_EUNSUPPORTED; }
  }
  const double cells = (double)(p->nx + 2 * p->ghostWidth) * (p->ny + 2 * p->ghostWidth) * (three_d ? p->nz + 2 * p->ghostWidth : 1);
  if (cells >= 4294967295.0) { *why = "more than 2^32 cells per device"; return RGPU_EUNSUPPORTED; }
  return RGPU_OK;
}

void fill_dev_params(const rgpu_params& p, DevParams* g) {
  std::memset(g, 0, sizeof(*g));
  g->three_d = (p.nz_global != 1) ? 1 : 0;
  g->gw = p.ghostWidth;
  g->nx = p.nx; g->ny = p.ny; g->nz = p.nz;
  g->isize = p.nx + 2 * p.ghostWidth;
  g->jsize = p.ny + 2 * p.ghostWidth;
  g->ksize = g->three_d ? p.nz + 2 * p.ghostWidth : 1;
  g->nvar = p.nbVar;
  g->mhd = p.mhdEnabled;
  g->rot = (p.mhdEnabled && p.Omega0 > 0) ? 1 : 0;
  g->shearbox = p.shearingBoxEnabled;
  g->sj = (unsigned)g->isize;
  g->sk = (unsigned)g->isize * (unsigned)g->jsize;
  g->ncell = (unsigned long long)g->isize * g->jsize * g->ksize;
  g->fsj = g->sj; g->fsk = g->sk; g->foff = 0; g->fN = g->ncell;
  if (g->three_d && g->mhd) {
    // F / emf of the 3D MHD step: rows of whole 128-byte lines (a multiple of 16 doubles), cell i = gw of every row on a line boundary,
    // so that the 16-cell row segments a wave of the sweep writes are whole lines.  Same-box A/B at 512^3 over seven boxes
    // (profiles/r06_flux_pitch_ab.txt): sweep -1.0 ... -1.2 ms in both builds, WRITE_SIZE 26.4 -> 20.6 GB, update level; rows of 520
    // doubles (64-byte granules only) gave the same sweep but an update 0.1 ... 1.3 ms slower, depending on the box.
    g->fsj = (((unsigned)g->isize + 15u) / 16u) * 16u;
    g->foff = (16u - (unsigned)g->gw % 16u) % 16u;
    g->fsk = g->fsj * (unsigned)g->jsize;
    g->fN = ((unsigned long long)g->fsk * g->ksize + g->foff + 15ull) & ~15ull;
  }
  g->dx = p.dx; g->dy = p.dy; g->dz = p.dz; g->xMin = p.xMin; g->deltaX = p.xMax - p.xMin;
  g->gamma0 = p.gamma0; g->cIso = p.cIso; g->smallr = p.smallr; g->smallc = p.smallc; g->smallp = p.smallp;
  g->smallpp = p.smallpp; g->gamma6 = p.gamma6; g->Omega0 = p.Omega0;
  g->slope_type = p.slope_type;
  g->mag_slope_type = std::fmin(p.slope_type, 2.0);
  g->niter_riemann = p.niter_riemann; g->riemannSolver = p.riemannSolver; g->magRiemannSolver = p.magRiemannSolver;
  g->dirwise_update = (!p.mhdEnabled && p.unsplitVersion == 2) ? 1 : 0; g->xcd_sub = 0;
  // interfaces INSIDE the global box only: the periodic wrap between the last and the first slab is a boundary of the
  // reference's single domain and keeps its ranges
  g->zlo_copy = (p.bc[4] == RGPU_BC_COPY && p.slab_rank > 0) ? 1 : 0;
  g->zhi_copy = (p.bc[5] == RGPU_BC_COPY && p.slab_rank < p.slab_count - 1) ? 1 : 0;
  g->grav_on = 0; g->hgx = 0.0; g->hgy = 0.0; g->hgz = 0.0; g->G = 0; g->hdt = 0.0;   // per step: step_core_planes
}

// number of scratch doubles per cell for each array of the active solver family
struct ScratchPlan { int q, e, t, f, emf; };
void fill_dev_params(const rgpu_params& p, DevParams* g);
ScratchPlan plan_for(const rgpu_params& p) {
  const bool three_d = p.nz_global != 1;
  ScratchPlan s;
  if (!p.mhdEnabled) {
    const int nv = three_d ? 5 : 4, nd = three_d ? 3 : 2;
    s.q = nv; s.e = 0; s.t = nv * (1 + nd); s.f = nv * nd; s.emf = 0;
  } else if (!three_d) {
    s.q = 8; s.e = 0; s.t = T2_COUNT; s.f = F2_COUNT; s.emf = 0;
  } else {
    s.q = 8; s.e = 3; s.t = T_COUNT; s.f = F_COUNT; s.emf = 3;
  }
  // The LDS-tiled sweeps keep primitives, electric field and traced state on chip: when the backend covers the run's
  // configuration those arrays are never touched and are not allocated (518^3 MHD: 38 instead of 92 GB of device memory).
  // F stays (hydro: scratch of the viscous fluxes and of the history sums); T keeps three components when the resistive
  // stage borrows it for its emf.
  DevParams g;
  fill_dev_params(p, &g);
  if (three_d && !p.mhdEnabled && rgpu_tiled::hydro3d_sweep_covers(g) && p.gravityEnabled != 2) { s.q = 0; s.t = 0; }
  if (three_d && p.mhdEnabled && rgpu_tiled::mhd3d_sweep_covers(g) && p.gravityEnabled != 2) { s.q = 0; s.e = 0; s.t = (p.eta > 0) ? 3 : 0; }
  return s;
}

int alloc_zero(rgpu_ctx* c, double** ptr, size_t doubles) {
  *ptr = 0;
  if (doubles == 0) return 0;
  if (rg_malloc((void**)ptr, doubles * sizeof(double))) return -1;
  c->scratch_bytes += doubles * sizeof(double);
  // zero once: cells outside a kernel's index range are never written but may be read by over-wide neighbours
  return rg_memset_async(*ptr, 0, doubles * sizeof(double), c->stream);
}

int create_common(const rgpu_params* p, double* dU, double* dU2, void* hip_stream, bool external, rgpu_ctx** out) {
  if (!out) return RGPU_EINVAL;
  *out = 0;
  std::string why;
  const int vr = validate(p, &why);
  rgpu_ctx* c = new (std::nothrow) rgpu_ctx();
  if (!c) return RGPU_ENOMEM;
  *out = c;  // returned even on failure so that rgpu_last_error can be read; caller destroys it
  std::memset(&c->p, 0, sizeof(c->p));
  if (p) c->p = *p;
  c->own_state = !external;
  c->U[0] = c->U[1] = 0;
  c->Q = c->E = c->T = c->F = c->emf = c->shear_save = c->shear_remap = 0;
  c->G = 0;
  c->Frc = 0;
  c->ou = 0;
  c->d_red = 0; c->d_red_base = 0; c->fold_mode = false; c->fold_request = false; c->fold_pending = false; c->fold_phase0 = 0; c->h_red = 0; c->d_clk = 0; c->h_clk = 0; c->clk_cur = 0; c->clk_n = -1; c->clk_t0 = 0.0; c->clk_tEnd = 0.0;
  c->scratch_bytes = 0;
  c->timers_on = false; c->ev_ok = false;
  for (int i = 0; i < RGPU_T_COUNT; ++i) { c->t_acc[i] = 0; c->t_calls[i] = 0; }
  c->stream = (rg_stream_t)0;
  c->stream2 = (rg_stream_t)0;
  c->nchunks = 1;
  c->n_order_events = 0; c->fork_ok = false;
  c->device = -1;
  c->xcd_sub = 4096;
  c->fused_dt_parity = -1;
  c->fused_dt_slots = 1;
  c->scan_acc_parity = -1;
  c->ghost_ok_parity = -1;
  if (vr) return fail(c, vr, why);
  if (rg_device_count() < 1) return fail(c, RGPU_ENODEVICE, "no HIP device: this library has no CPU fallback (backend " RG_BACKEND_NAME ")");
  c->device = rg_current_device();
  if (external) {   // adopted arrays must live on the device the context will launch on
    const int d1 = rg_pointer_device(dU), d2 = rg_pointer_device(dU2);
    if (dU && dU2 && d1 >= 0 && d2 >= 0) {
      if (d1 != d2) return fail(c, RGPU_EINVAL, "external state arrays live on different devices");
      c->device = d1;
      rg_set_device(d1);
    }
  }
  fill_dev_params(*p, &c->g);
  c->ncell = (size_t)c->g.ncell;
  c->n32 = (unsigned)c->ncell;
  if (external) {
    if (!dU || !dU2) return fail(c, RGPU_EINVAL, "external state pointers are NULL");
    c->U[0] = dU; c->U[1] = dU2;
    c->stream = rg_stream_from_handle(hip_stream);
  } else {
    const size_t n = c->ncell * (size_t)p->nbVar;
    if (alloc_zero(c, &c->U[0], n) || alloc_zero(c, &c->U[1], n)) return fail(c, RGPU_ENOMEM, "device allocation of the state arrays failed");
  }
  const ScratchPlan sp = plan_for(*p);
  if (p->randomForcingEnabled && alloc_zero(c, &c->Frc, c->ncell * 3)) return fail(c, RGPU_ENOMEM, "device allocation of the forcing field failed");
  if (p->ouForcingEnabled) {   // == init_forcing() of the reference's init_hydro_turbulence_Ornstein_Uhlenbeck (HydroRunBase.cpp:6990)
    c->ou = new (std::nothrow) rgpu_ou::OuProcess();
    if (!c->ou) return fail(c, RGPU_ENOMEM, "allocation of the forcing process failed");
    c->ou->init(p->ouInitRandom, p->ouTimeScaleTurb, p->ouAmplitudeTurb, p->ouKsi);
  }
  if (p->gravityEnabled == 2 && alloc_zero(c, &c->G, c->ncell * 3)) return fail(c, RGPU_ENOMEM, "device allocation of the gravity field failed");
  if (alloc_zero(c, &c->Q, c->ncell * sp.q) || alloc_zero(c, &c->E, c->ncell * sp.e) || alloc_zero(c, &c->T, c->ncell * sp.t) ||
      alloc_zero(c, &c->F, (size_t)c->g.fN * sp.f) || alloc_zero(c, &c->emf, (size_t)c->g.fN * sp.emf))
    return fail(c, RGPU_ENOMEM, "device allocation of the scratch arrays failed");
  if (c->g.shearbox) {
    const size_t P = (size_t)c->g.jsize * c->g.ksize;
    if (alloc_zero(c, &c->shear_save, 2 * P) || alloc_zero(c, &c->shear_remap, 2 * P))
      return fail(c, RGPU_ENOMEM, "device allocation of the shear buffers failed");
  }
  static_assert((int)RG_DT_SLOTS == RGPU_DT_SLOTS, "include/rgpu.h promises RGPU_DT_SLOTS device slots");
  if (rg_malloc((void**)&c->d_red_base, 3 * RG_DT_SLOTS * sizeof(unsigned long long)) || rg_host_alloc((void**)&c->h_red, RG_DT_SLOTS * sizeof(unsigned long long)) ||
      rg_memset_async(c->d_red_base, 0, 3 * RG_DT_SLOTS * sizeof(unsigned long long), c->stream))
    return fail(c, RGPU_ENOMEM, "allocation of the reduction slots failed");
  c->d_red = c->d_red_base;
  if (rg_event_create(&c->ev0) == 0 && rg_event_create(&c->ev1) == 0) c->ev_ok = true;
  c->nchunks = 1;
  // sub-band size (cells) of the XCD-aware workgroup order, 0 = linear order (rg_backend.h: rg_launch_planes)
  if (rgpu::options().xcd_sub >= 0) c->xcd_sub = (unsigned)rgpu::options().xcd_sub;
  c->g.xcd_sub = (int)c->xcd_sub;
  if (p->mhdEnabled && c->g.three_d) {
    // (the flat kernels only -- RGPU_TILED=0 or a per-cell gravity field; the tiled sweep marches z inside one launch)
    // default: chunks of ~8 planes (measured best at 512^3: 64 chunks 75.7 ms/step vs 82-84 ms serial; 128 chunks
    // 77.4, 256 chunks 82.6); option "chunks" = 1 selects the serial single-stream schedule.  Equal stream priorities
    // (a low-priority VALU stream measured 3 % slower).
    int want = rgpu::options().chunks > 0 ? rgpu::options().chunks : c->g.ksize / 8;
    if (want > c->g.ksize / 2) want = c->g.ksize / 2;
    if (want > rgpu_ctx::kMaxChunks) want = rgpu_ctx::kMaxChunks;
    if (want > 1 && rg_stream_create(&c->stream2, 0) == 0) {
      bool ok = c->fork_ok = rg_order_event_create(&c->ev_fork) == 0;
      for (int i = 0; i < want && ok; ++i) {
        if (rg_order_event_create(&c->ev_trace[i])) { ok = false; break; }
        if (rg_order_event_create(&c->ev_flux[i])) { rg_event_destroy(c->ev_trace[i]); ok = false; break; }
        c->n_order_events = i + 1;
      }
      if (ok) c->nchunks = want;
    }
  }
  if (rg_stream_sync(c->stream)) return fail(c, RGPU_EHIP, std::string("device error during creation: ") + rg_last_error_string());
  return RGPU_OK;
}

}  // namespace
