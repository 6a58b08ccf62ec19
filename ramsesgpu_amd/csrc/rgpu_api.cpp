// rgpu_api.cpp -- implementation of the C ABI of include/rgpu.h: context, step driver, ghost fill, CFL scan.
//
// Compiled by hipcc (-x hip --offload-arch=gfx950 -ffp-contract=off).  All work of a context is issued on one
// HIP stream in program order; the only host<->device traffic inside the path is the 8-byte result of the CFL
// reduction (the reference copies <=192 partial maxima per step, MHDRunBase.cpp:103-128).
#include "../../include/rgpu.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

#include "launchers.h"
#include "rg_options.h"
#include "rg_tiled.h"   // LDS-tiled cooperative kernels of the backend (hip/rg_tiled.h)

using namespace rgpu;
using namespace rgpu_dev;

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
  c->d_red = 0; c->d_red_base = 0; c->fold_mode = false; c->fold_pending = false; c->fold_phase0 = 0; c->h_red = 0; c->d_clk = 0; c->h_clk = 0; c->clk_cur = 0; c->clk_n = -1; c->clk_t0 = 0.0; c->clk_tEnd = 0.0;
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
      alloc_zero(c, &c->F, c->ncell * sp.f) || alloc_zero(c, &c->emf, c->ncell * sp.emf))
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
  return !c->g.three_d || (c->p.bc[4] != RGPU_BC_Z_STRATIFIED && c->p.bc[5] != RGPU_BC_Z_STRATIFIED);
}

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

// ---- history diagnostics ------------------------------------------------------------------------------------------
// Scratch lives in the flux array F, which is dead between steps: rows [HIST_NQ][nk][isize], then the column sums
// [HIST_NQ][isize], then the two mean-velocity columns.
struct HistScratch { double* rows; double* cols; double* mean; size_t R; };
HistScratch hist_scratch(rgpu_ctx* c) {
  HistScratch h;
  const int nk = c->g.three_d ? c->g.nz : 1;
  h.R = (size_t)c->g.isize * nk;
  h.rows = c->F;
  h.cols = c->F + (size_t)HIST_NQ * h.R;
  h.mean = h.cols + (size_t)HIST_NQ * c->g.isize;
  return h;
}

int history_columns(rgpu_ctx* c, int parity, double* h_cols) {
  const HistScratch h = hist_scratch(c);
  K_hist_rows kr = {c->g, c->U[parity & 1], h.rows};
  K_hist_cols kc = {c->g, h.rows, h.cols, HIST_NQ};
  if (rg_launch<kBlock>(c->stream, (unsigned)h.R, kr) || rg_launch<kBlock>(c->stream, (unsigned)(HIST_NQ * c->g.isize), kc)) return -1;
  if (rg_copy_d2h(h_cols, h.cols, sizeof(double) * HIST_NQ * c->g.isize, c->stream) || rg_stream_sync(c->stream)) return -1;
  return 0;
}

int history_reynolds(rgpu_ctx* c, int parity, const double* h_mean_vx, const double* h_mean_vy, double dTau, double* h_cols) {
  const HistScratch h = hist_scratch(c);
  const size_t is = (size_t)c->g.isize;
  if (rg_copy_h2d(h.mean, h_mean_vx, sizeof(double) * is, c->stream) || rg_copy_h2d(h.mean + is, h_mean_vy, sizeof(double) * is, c->stream)) return -1;
  K_hist_reynolds kr = {c->g, c->U[parity & 1], h.mean, h.mean + is, dTau, h.rows};
  K_hist_cols kc = {c->g, h.rows, h.cols, 1};
  if (rg_launch<kBlock>(c->stream, (unsigned)h.R, kr) || rg_launch<kBlock>(c->stream, (unsigned)is, kc)) return -1;
  if (rg_copy_d2h(h_cols, h.cols, sizeof(double) * is, c->stream) || rg_stream_sync(c->stream)) return -1;
  return 0;
}

// random forcing: the two sums of compute_random_forcing_normalization over this domain's interior, reduced in the
// rows (along y) / columns (along z) / host (along x) order of the history sums
int forcing_sums(rgpu_ctx* c, int parity, double* out2) {
  const HistScratch h = hist_scratch(c);
  const size_t is = (size_t)c->g.isize;
  K_forcing_rows kr = {c->g, c->U[parity & 1], c->Frc, h.rows};
  K_hist_cols kc = {c->g, h.rows, h.cols, 2};
  if (rg_launch<kBlock>(c->stream, (unsigned)h.R, kr) || rg_launch<kBlock>(c->stream, (unsigned)(2 * is), kc)) return -1;
  std::vector<double> cols(2 * is);
  if (rg_copy_d2h(cols.data(), h.cols, sizeof(double) * 2 * is, c->stream) || rg_stream_sync(c->stream)) return -1;
  out2[0] = 0.0; out2[1] = 0.0;
  for (size_t i = 0; i < is; ++i) { out2[0] += cols[i]; out2[1] += cols[is + i]; }
  return 0;
}

double forcing_norm(const rgpu_params& p, const double* s, double dt) {   // HydroRunBase.cpp:1286-1293
  if (p.randomForcingEdot == 0) return 0.0;
  const long long nbCells = (long long)p.nx * p.ny * p.nz_global;
  return (std::sqrt(s[0] * s[0] + s[1] * dt * p.randomForcingEdot * 2 * nbCells) - s[0]) / s[1];
}

int add_forcing(rgpu_ctx* c, int parity, double norm) {
  state_modified(c);
  K_add_forcing k = {c->g, c->U[parity & 1], c->Frc, norm};
  return launch_planes<kBlock, 1>(c->stream, c->g, clip(c->g.gw, c->g.ksize - c->g.gw, c->g.ksize), k);
}

// Ornstein-Uhlenbeck forcing on U[parity]: advance the modes on the host, then one kernel over the interior planes
int step_ou_forcing(rgpu_ctx* c, int parity, double dt) {
  if (!c->ou) return 0;
  state_modified(c);
  Phase ph(c, RGPU_T_UPDATE);
  c->ou->update(dt, c->p.cIso);
  K_ou_forcing k = {c->g, c->U[parity & 1], c->ou->m, dt, c->p.yMin, c->p.zMin, c->p.slab_rank * c->p.nz};
  return launch_planes<kBlock, 1>(c->stream, c->g, clip(c->g.gw, c->g.ksize - c->g.gw, c->g.ksize), k);
}

int step_forcing(rgpu_ctx* c, int nStep, double dt) {
  if (!c->p.randomForcingEnabled) return 0;
  Phase ph(c, RGPU_T_UPDATE);
  double s[2];
  if (forcing_sums(c, (nStep + 1) % 2, s)) return -1;
  return add_forcing(c, (nStep + 1) % 2, forcing_norm(c->p, s, dt));
}

// every entry point makes the context's device current: a multi-GPU process (or a thread whose current device differs)
// would otherwise launch on the wrong device
struct K_selftest_arith {
  const double* num; const double* den; double* quot; double* quot2; double* root; double* root2;
  RG_DEVFN void operator()(unsigned i) const {
    quot[i] = rg_div(num[i], rg_recip(den[i]));
    quot2[i] = num[i] / den[i];
    root[i] = rg_sqrt(num[i]);
    root2[i] = sqrt(num[i]);
  }
};

// one sample = the four corner states of an edge (LL, RL, LR, RR: r p u v w a b c each) and their four electric fields, SoA: in[q * n + i]
struct K_selftest_alfven {
  DevParams g; const double* in; double* e_sel; double* e_ref; int* route; unsigned n;
  RG_DEVFN void operator()(unsigned i) const {
    Prim8 s[4];
    for (int q = 0; q < 4; ++q) {
      const double* x = in + (size_t)(8 * q) * n + i;
      s[q].r = x[0]; s[q].p = x[n]; s[q].u = x[2 * (size_t)n]; s[q].v = x[3 * (size_t)n]; s[q].w = x[4 * (size_t)n];
      s[q].a = x[5 * (size_t)n]; s[q].b = x[6 * (size_t)n]; s[q].c = x[7 * (size_t)n];
    }
    const double E0 = in[(size_t)32 * n + i], E1 = in[(size_t)33 * n + i], E2 = in[(size_t)34 * n + i], E3 = in[(size_t)35 * n + i];
    int r = 0;
    e_sel[i] = mag_hlld_2d<false>(g, s[0], s[1], s[2], s[3], E0, E1, E2, E3, &r);
    e_ref[i] = mag_hlld_2d<true>(g, s[0], s[1], s[2], s[3], E0, E1, E2, E3);
    route[i] = r;
  }
};

#define RG_CHECK_CTX(c) do { if (!(c)) return RGPU_EINVAL; if ((c)->device >= 0) rg_set_device((c)->device); } while (0)
#define RG_HIPFAIL(c, what) fail((c), RGPU_EHIP, std::string(what) + ": " + rg_last_error_string())

}  // namespace

// =================================================================================================================
extern "C" {

int rgpu_create(const rgpu_params* p, rgpu_ctx** out) { return create_common(p, 0, 0, 0, false, out); }

int rgpu_create_external(const rgpu_params* p, double* dU, double* dU2, void* hip_stream, rgpu_ctx** out) {
  return create_common(p, dU, dU2, hip_stream, true, out);
}

void rgpu_destroy(rgpu_ctx* c) {
  if (!c) return;
  if (c->device >= 0) rg_set_device(c->device);
  if (c->own_state) { rg_free(c->U[0]); rg_free(c->U[1]); }
  rg_free(c->Q); rg_free(c->E); rg_free(c->T); rg_free(c->F); rg_free(c->emf); rg_free(c->shear_save); rg_free(c->shear_remap); rg_free(c->G); rg_free(c->Frc);
  delete c->ou;
  rg_free(c->d_red_base); rg_host_free(c->h_red);
  if (c->d_clk) rg_free(c->d_clk);
  if (c->h_clk) rg_host_free(c->h_clk);
  if (c->ev_ok) { rg_event_destroy(c->ev0); rg_event_destroy(c->ev1); }
  if (c->fork_ok) rg_event_destroy(c->ev_fork);
  for (int i = 0; i < c->n_order_events; ++i) { rg_event_destroy(c->ev_trace[i]); rg_event_destroy(c->ev_flux[i]); }
  if (c->stream2) rg_stream_destroy(c->stream2);
  delete c;
}

size_t rgpu_device_bytes(const rgpu_params* p) {
  if (!p) return 0;
  const bool three_d = p->nz_global != 1;
  const size_t isize = p->nx + 2 * p->ghostWidth, jsize = p->ny + 2 * p->ghostWidth, ksize = three_d ? p->nz + 2 * p->ghostWidth : 1;
  const size_t ncell = isize * jsize * ksize;
  const ScratchPlan sp = plan_for(*p);
  size_t doubles = ncell * (size_t)(2 * p->nbVar + sp.q + sp.e + sp.t + sp.f + sp.emf);
  if (p->shearingBoxEnabled) doubles += 4 * jsize * ksize;
  if (p->gravityEnabled == 2) doubles += 3 * ncell;
  if (p->randomForcingEnabled) doubles += 3 * ncell;
  return doubles * sizeof(double);
}

const char* rgpu_last_error(rgpu_ctx* c) { return c ? c->err.c_str() : "null context"; }

int rgpu_upload(rgpu_ctx* c, const double* hU, int both) {
  RG_CHECK_CTX(c);
  state_modified(c);
  if (!hU || !c->U[0]) return fail(c, RGPU_EINVAL, "upload: null pointer / context without state");
  const size_t bytes = c->ncell * (size_t)c->p.nbVar * sizeof(double);
  if (rg_copy_h2d(c->U[0], hU, bytes, c->stream)) return RG_HIPFAIL(c, "upload");
  if (both && rg_copy_d2d(c->U[1], c->U[0], bytes, c->stream)) return RG_HIPFAIL(c, "upload (copy to U2)");
  if (rg_stream_sync(c->stream)) return RG_HIPFAIL(c, "upload sync");
  return RGPU_OK;
}

int rgpu_set_gravity_field(rgpu_ctx* c, const double* hG) {
  RG_CHECK_CTX(c);
  if (!hG) return fail(c, RGPU_EINVAL, "set_gravity_field: null pointer");
  if (c->p.gravityEnabled != 2 || !c->G) return fail(c, RGPU_EINVAL, "set_gravity_field: the context was not created with gravityEnabled = 2");
  if (rg_copy_h2d(c->G, hG, c->ncell * 3 * sizeof(double), c->stream) || rg_stream_sync(c->stream)) return RG_HIPFAIL(c, "set_gravity_field");
  return RGPU_OK;
}

int rgpu_set_forcing_field(rgpu_ctx* c, const double* hF) {
  RG_CHECK_CTX(c);
  if (!hF) return fail(c, RGPU_EINVAL, "set_forcing_field: null pointer");
  if (!c->p.randomForcingEnabled || !c->Frc) return fail(c, RGPU_EINVAL, "set_forcing_field: the context was not created with randomForcingEnabled");
  if (rg_copy_h2d(c->Frc, hF, c->ncell * 3 * sizeof(double), c->stream) || rg_stream_sync(c->stream)) return RG_HIPFAIL(c, "set_forcing_field");
  return RGPU_OK;
}

int rgpu_step_ou_forcing(rgpu_ctx* c, int parity, double dt) {
  RG_CHECK_CTX(c);
  if (!c->ou) return fail(c, RGPU_EINVAL, "step_ou_forcing: the context was not created with ouForcingEnabled");
  if (step_ou_forcing(c, parity, dt)) return RG_HIPFAIL(c, "step_ou_forcing");
  return RGPU_OK;
}

int rgpu_ou_forcing_state(rgpu_ctx* c, double* mode93, double* forcingField93) {
  RG_CHECK_CTX(c);
  if (!c->ou || !mode93 || !forcingField93) return fail(c, RGPU_EINVAL, "ou_forcing_state: no forcing process / null pointer");
  std::memcpy(mode93, c->ou->m.mode, sizeof(c->ou->m.mode));
  std::memcpy(forcingField93, c->ou->m.force, sizeof(c->ou->m.force));
  return RGPU_OK;
}

static_assert(RGPU_OU_STATE_DOUBLES == rgpu_ou::OuProcess::STATE_DOUBLES, "rgpu.h out of sync with ou_forcing.h");
int rgpu_ou_forcing_get_state(rgpu_ctx* c, double* state) {
  RG_CHECK_CTX(c);
  if (!c->ou || !state) return fail(c, RGPU_EINVAL, "ou_forcing_get_state: no forcing process / null pointer");
  c->ou->get_state(state);
  return RGPU_OK;
}
int rgpu_ou_forcing_set_state(rgpu_ctx* c, const double* state) {
  RG_CHECK_CTX(c);
  if (!c->ou || !state) return fail(c, RGPU_EINVAL, "ou_forcing_set_state: no forcing process / null pointer");
  c->ou->set_state(state);
  return RGPU_OK;
}

int rgpu_forcing_sums(rgpu_ctx* c, int parity, double* out) {
  RG_CHECK_CTX(c);
  if (!out || !c->Frc) return fail(c, RGPU_EINVAL, "forcing_sums: null pointer / context without forcing field");
  if (forcing_sums(c, parity, out)) return RG_HIPFAIL(c, "forcing_sums");
  return RGPU_OK;
}

int rgpu_add_forcing(rgpu_ctx* c, int parity, double norm) {
  RG_CHECK_CTX(c);
  if (!c->Frc) return fail(c, RGPU_EINVAL, "add_forcing: context without forcing field");
  if (add_forcing(c, parity, norm) || rg_stream_sync(c->stream)) return RG_HIPFAIL(c, "add_forcing");
  return RGPU_OK;
}

int rgpu_download(rgpu_ctx* c, double* hU, int parity) {
  RG_CHECK_CTX(c);
  if (!hU || !c->U[0]) return fail(c, RGPU_EINVAL, "download: null pointer / context without state");
  const size_t bytes = c->ncell * (size_t)c->p.nbVar * sizeof(double);
  if (rg_copy_d2h(hU, c->U[parity & 1], bytes, c->stream) || rg_stream_sync(c->stream)) return RG_HIPFAIL(c, "download");
  return RGPU_OK;
}

double* rgpu_device_state(rgpu_ctx* c, int parity) { return c ? c->U[parity & 1] : 0; }
int rgpu_get_params(rgpu_ctx* c, rgpu_params* out) { if (!c || !out) return RGPU_EINVAL; *out = c->p; return RGPU_OK; }
void* rgpu_stream_handle(rgpu_ctx* c) { return c ? rg_stream_to_handle(c->stream) : 0; }
double* rgpu_inv_dt_device_slot(rgpu_ctx* c) { return c ? reinterpret_cast<double*>(c->d_red) : 0; }

int rgpu_read_cell(rgpu_ctx* c, int parity, int i, int j, int k, double* out) {
  RG_CHECK_CTX(c);
  if (!out || !c->U[0]) return fail(c, RGPU_EINVAL, "read_cell: null pointer / context without state");
  const DevParams& g = c->g;
  if (i < 0 || i >= g.isize || j < 0 || j >= g.jsize || k < 0 || k >= g.ksize) return fail(c, RGPU_EINVAL, "read_cell: index outside the array");
  const size_t idx = (size_t)i + (size_t)g.isize * ((size_t)j + (size_t)g.jsize * (size_t)k);
  for (int v = 0; v < c->p.nbVar; ++v)
    if (rg_copy_d2h(out + v, c->U[parity & 1] + idx + (size_t)v * c->ncell, sizeof(double), c->stream)) return RG_HIPFAIL(c, "read_cell");
  if (rg_stream_sync(c->stream)) return RG_HIPFAIL(c, "read_cell");
  return RGPU_OK;
}

// A ghost fill called from OUTSIDE the step may change what the CFL scan reads: with a non-periodic MHD face it overwrites
// the field the CT update left on the first high ghost face (compute_dt_mhd reads it as the high-face field of the last
// interior cell).  The 1/dt a fused scan left in the device slot is then stale: drop it, the next compute_dt scans again.
// Periodic / copy / shearing faces rewrite ghosts with bit-identical images of interior values (or leave that face
// alone), so the scan result stands.
static void boundary_call_invalidates_dt(rgpu_ctx* c, int parity, int dim_lo, int dim_hi) {
  if (c->fused_dt_parity != (parity & 1)) return;   // (a scan being accumulated piece by piece belongs to the slab driver's own schedule)
  bool keeps = true;
  for (int d = dim_lo; d <= dim_hi; ++d) {
    if (d == RGPU_ZDIR && !c->g.three_d) continue;
    for (int side = 0; side < 2; ++side) {
      const int bc = c->p.bc[2 * (d - 1) + side];
      if (bc != RGPU_BC_PERIODIC && bc != RGPU_BC_COPY && bc != RGPU_BC_SHEARINGBOX) keeps = false;
    }
  }
  if (c->p.enableJet) keeps = false;
  if (!keeps) c->fused_dt_parity = -1;
}

int rgpu_invalidate_dt(rgpu_ctx* c) {
  if (!c) return RGPU_EINVAL;
  state_modified(c);
  return RGPU_OK;
}

int rgpu_make_boundaries(rgpu_ctx* c, int parity, int idim) {
  RG_CHECK_CTX(c);
  if (!c->U[0]) return fail(c, RGPU_EINVAL, "context was not created");
  if (idim < RGPU_XDIR || idim > RGPU_ZDIR) return fail(c, RGPU_EINVAL, "idim must be 1,2,3");
  boundary_call_invalidates_dt(c, parity, idim, idim);
  Phase ph(c, RGPU_T_BOUNDARIES);
  if (do_make_boundaries(c, c->U[parity & 1], idim)) return RG_HIPFAIL(c, "make_boundaries");
  return RGPU_OK;
}

int rgpu_make_boundaries_shear(rgpu_ctx* c, int parity, double totalTime, double dt) {
  RG_CHECK_CTX(c);
  if (!c->U[0]) return fail(c, RGPU_EINVAL, "context was not created");
  if (!(c->g.shearbox && c->g.three_d)) return fail(c, RGPU_EINVAL, "shearing box is not enabled");
  boundary_call_invalidates_dt(c, parity, RGPU_XDIR, RGPU_XDIR);
  Phase ph(c, RGPU_T_BOUNDARIES);
  if (do_make_boundaries_shear(c, c->U[parity & 1], totalTime, dt)) return RG_HIPFAIL(c, "make_boundaries_shear");
  return RGPU_OK;
}

int rgpu_make_all_boundaries(rgpu_ctx* c, int parity, double totalTime, double dt) {
  RG_CHECK_CTX(c);
  if (!c->U[0]) return fail(c, RGPU_EINVAL, "context was not created");
  boundary_call_invalidates_dt(c, parity, RGPU_XDIR, RGPU_ZDIR);
  Phase ph(c, RGPU_T_BOUNDARIES);
  double* U = c->U[parity & 1];
  int rc;
  if (c->g.shearbox && c->g.three_d) {
    rc = do_make_boundaries(c, U, RGPU_YDIR) || do_make_boundaries_shear(c, U, totalTime, dt) ||
         do_make_boundaries(c, U, RGPU_ZDIR) || do_make_boundaries(c, U, RGPU_YDIR);
  } else {
    rc = do_make_boundaries(c, U, RGPU_XDIR) || do_make_boundaries(c, U, RGPU_YDIR) ||
         (c->g.three_d && do_make_boundaries(c, U, RGPU_ZDIR));
  }
  if (rc) return RG_HIPFAIL(c, "make_all_boundaries");
  return RGPU_OK;
}

int rgpu_compute_inv_dt(rgpu_ctx* c, int parity, double* invDt) {
  RG_CHECK_CTX(c);
  if (!invDt || !c->U[0]) return fail(c, RGPU_EINVAL, "compute_inv_dt: null pointer / context without state");
  if (inv_dt(c, parity, invDt)) return RG_HIPFAIL(c, "compute_inv_dt");
  return RGPU_OK;
}

int rgpu_inv_dt_accumulate(rgpu_ctx* c, int parity, int k_lo, int k_hi, int reset) {
  RG_CHECK_CTX(c);
  if (!c->U[0]) return fail(c, RGPU_EINVAL, "context was not created");
  if (!c->g.three_d) return fail(c, RGPU_EINVAL, "inv_dt_accumulate: plane ranges need a 3D context");
  if (k_lo < 0) k_lo = 0;
  if (k_hi > c->g.ksize) k_hi = c->g.ksize;
  if (k_hi < k_lo) k_hi = k_lo;
  if (inv_dt_scan(c, parity, (unsigned)k_lo * c->g.sk, (unsigned)(k_hi - k_lo) * c->g.sk, reset != 0)) return RG_HIPFAIL(c, "inv_dt_accumulate");
  return RGPU_OK;
}

int rgpu_inv_dt_result(rgpu_ctx* c, double* invDt) {
  RG_CHECK_CTX(c);
  if (!invDt || !c->U[0]) return fail(c, RGPU_EINVAL, "inv_dt_result: null pointer / context without state");
  // slab contexts: always every slot -- the ranks all-reduce a fixed RG_DT_SLOTS values, and a rank after a full scan (slot 0 + zeros,
  // inv_dt_scan) must still see a peer's fused maxima in the other slots
  const int nslots = (c->p.slab_count > 1) ? (int)RG_DT_SLOTS : (c->fused_dt_parity >= 0 ? c->fused_dt_slots : 1);
  if (inv_dt_fetch(c, invDt, nslots)) return RG_HIPFAIL(c, "inv_dt_result");
  return RGPU_OK;
}
int rgpu_inv_dt_fusable(rgpu_ctx* c) {
  if (!c || !c->U[0] || !c->g.three_d) return 0;
  return (c->p.mhdEnabled ? mhd3d_scan_cond(c) : hydro3d_scan_cond(c)) ? 1 : 0;
}
int rgpu_inv_dt_fused_active(rgpu_ctx* c, int parity) { return (c && c->U[0] && c->scan_acc_parity == (parity & 1)) ? 1 : 0; }
int rgpu_inv_dt_fused_commit(rgpu_ctx* c, int parity) {
  if (!c || !c->U[0]) return 0;
  if (c->scan_acc_parity != (parity & 1)) { c->scan_acc_parity = -1; return 0; }
  c->scan_acc_parity = -1;
  c->fused_dt_parity = parity & 1;
  c->fused_dt_slots = RG_DT_SLOTS;
  return RG_DT_SLOTS;
}

int rgpu_history_columns(rgpu_ctx* c, int parity, double* cols) {
  RG_CHECK_CTX(c);
  if (!cols || !c->U[0]) return fail(c, RGPU_EINVAL, "history_columns: null pointer / context without state");
  if (!c->p.mhdEnabled) return fail(c, RGPU_EUNSUPPORTED, "history diagnostics are defined for MHD runs");
  if (history_columns(c, parity, cols)) return RG_HIPFAIL(c, "history_columns");
  return RGPU_OK;
}

int rgpu_history_reynolds(rgpu_ctx* c, int parity, const double* mean_vx, const double* mean_vy, double dTau, double* cols) {
  RG_CHECK_CTX(c);
  if (!cols || !mean_vx || !mean_vy || !c->U[0]) return fail(c, RGPU_EINVAL, "history_reynolds: null pointer / context without state");
  if (!c->p.mhdEnabled) return fail(c, RGPU_EUNSUPPORTED, "history diagnostics are defined for MHD runs");
  if (history_reynolds(c, parity, mean_vx, mean_vy, dTau, cols)) return RG_HIPFAIL(c, "history_reynolds");
  return RGPU_OK;
}

int rgpu_history_mri(rgpu_ctx* c, int parity, double* out) {
  RG_CHECK_CTX(c);
  if (!out || !c->U[0]) return fail(c, RGPU_EINVAL, "history_mri: null pointer / context without state");
  if (!c->p.mhdEnabled) return fail(c, RGPU_EUNSUPPORTED, "history diagnostics are defined for MHD runs");
  if (c->p.slab_count > 1) return fail(c, RGPU_EINVAL, "slab contexts: combine rgpu_history_columns / _reynolds across ranks");
  const rgpu_params& p = c->p;
  const int is = c->g.isize, gw = c->g.gw;
  std::vector<double> cols((size_t)HIST_NQ * is), rcol(is), mvx(is), mvy(is);
  if (history_columns(c, parity, cols.data())) return RG_HIPFAIL(c, "history_mri");
  double dTau = p.dx * p.dy;
  if (c->g.three_d) dTau = p.dx * p.dy * p.dz / (p.xMax - p.xMin) / (p.yMax - p.yMin) / (p.zMax - p.zMin);   // MHDRunBase.cpp:3533-3536
  else dTau = p.dx * p.dy / (p.xMax - p.xMin) / (p.yMax - p.yMin);                                         // :3351-3353
  const int nyz = p.ny * (c->g.three_d ? p.nz : 1);
  for (int i = 0; i < is; ++i) { mvx[i] = cols[(size_t)1 * is + i] / nyz; mvy[i] = cols[(size_t)2 * is + i] / nyz; }
  if (history_reynolds(c, parity, mvx.data(), mvy.data(), dTau, rcol.data())) return RG_HIPFAIL(c, "history_mri");
  double sum[HIST_NQ], reyn = 0.0;
  for (int q = 0; q < HIST_NQ; ++q) { sum[q] = 0.0; for (int i = gw; i < is - gw; ++i) sum[q] += cols[(size_t)q * is + i]; }
  for (int i = gw; i < is - gw; ++i) reyn += rcol[i];
  out[0] = sum[0] * dTau;         // mass
  out[1] = sum[4] * dTau;         // maxwell
  out[2] = reyn;                  // reynolds (dTau is inside the sum, as in the reference)
  out[3] = sum[3] * dTau / 2.;    // magp
  out[4] = sum[5] * dTau; out[5] = sum[6] * dTau; out[6] = sum[7] * dTau;   // mean B
  out[7] = sum[8];                // divB
  return RGPU_OK;
}

// the 18 raw sums of history_turbulence over the interior cells of THIS context (a slab: its own planes): 0 rho, 1 rho v^2,
// 2 v^2, 3 B^2, 4 m.B / sqrt(rho), 5-7 B, 8-10 m, 11-16 the DFT sums of Bx (local plane index in the z term), 17 div B
int rgpu_history_turbulence_sums(rgpu_ctx* c, int parity, double* s) {
  RG_CHECK_CTX(c);
  if (!s || !c->U[0]) return fail(c, RGPU_EINVAL, "history_turbulence: null pointer / context without state");
  if (!c->p.mhdEnabled || !c->g.three_d) return fail(c, RGPU_EUNSUPPORTED, "history_turbulence is defined for 3D MHD runs (it does nothing in 2D)");
  const int is = c->g.isize, gw = c->g.gw;
  // rows [NQ][nz][isize] and columns [NQ][isize] in the flux array, dead between steps (F has 15 components per cell)
  const size_t R = (size_t)is * c->g.nz;
  double* rows = c->F;
  double* cols = c->F + (size_t)HIST_TURB_NQ * R;
  K_hist_turb_rows kr = {c->g, c->U[parity & 1], rows};
  K_hist_cols kc = {c->g, rows, cols, HIST_TURB_NQ};
  std::vector<double> h((size_t)HIST_TURB_NQ * is);
  if (rg_launch<kBlock>(c->stream, (unsigned)R, kr) || rg_launch<kBlock>(c->stream, (unsigned)(HIST_TURB_NQ * is), kc) ||
      rg_copy_d2h(h.data(), cols, sizeof(double) * h.size(), c->stream) || rg_stream_sync(c->stream)) return RG_HIPFAIL(c, "history_turbulence");
  for (int q = 0; q < HIST_TURB_NQ; ++q) { s[q] = 0.0; for (int i = gw; i < is - gw; ++i) s[q] += h[(size_t)q * is + i]; }
  return RGPU_OK;
}

int rgpu_history_turbulence(rgpu_ctx* c, int parity, double* out) {
  RG_CHECK_CTX(c);
  if (!out) return fail(c, RGPU_EINVAL, "history_turbulence: null pointer");
  if (c->p.slab_count > 1) return fail(c, RGPU_EINVAL, "history_turbulence: single-domain contexts only (slabs: rgpu_comm_history_turbulence)");
  double s[HIST_TURB_NQ];
  if (const int rc = rgpu_history_turbulence_sums(c, parity, s)) return rc;
  const rgpu_params& p = c->p;
  const double dTau = p.dx * p.dy * p.dz / (p.xMax - p.xMin) / (p.yMax - p.yMin) / (p.zMax - p.zMin);
  const double pi = 2 * std::asin(1.0);
  const double mass = s[0] * dTau, eKin = s[1] * dTau, mean_v2 = s[2] * dTau, eMag = s[3] * dTau, helicity = s[4] * dTau;
  const double mBx = s[5] * dTau, mBy = s[6] * dTau, mBz = s[7] * dTau;
  const double mean_B = std::sqrt(mBx * mBx + mBy * mBy + mBz * mBz);
  const double mean_rho = s[0] * dTau;
  out[0] = mass; out[1] = s[17]; out[2] = eKin; out[3] = eMag; out[4] = helicity; out[5] = mean_rho; out[6] = mean_B;
  out[7] = mBx; out[8] = mBy; out[9] = mBz; out[10] = s[8] * dTau; out[11] = s[9] * dTau; out[12] = s[10] * dTau;
  out[13] = std::sqrt(mean_v2) / p.cIso;                                        // Ma_s
  out[14] = std::sqrt(mean_v2) / (mean_B / std::sqrt(4 * pi * mean_rho));       // Ma_alfven
  out[15] = std::sqrt(s[11] * s[11] + s[12] * s[12]) * dTau;
  out[16] = std::sqrt(s[13] * s[13] + s[14] * s[14]) * dTau;
  out[17] = std::sqrt(s[15] * s[15] + s[16] * s[16]) * dTau;
  return RGPU_OK;
}

int rgpu_state_checksum(rgpu_ctx* c, int parity, unsigned long long* out) {
  RG_CHECK_CTX(c);
  if (!out || !c->U[0]) return fail(c, RGPU_EINVAL, "state_checksum: null pointer / context without state");
  const size_t R = (size_t)c->g.nx * (c->g.three_d ? c->g.nz : 1);
  if (!c->F || R > c->ncell) return fail(c, RGPU_EINVAL, "state_checksum: no scratch for the row sums");
  // row sums in the flux array, dead between steps (as the history sums)
  unsigned long long* rows = reinterpret_cast<unsigned long long*>(c->F);
  K_checksum_rows k = {c->g, c->U[parity & 1], rows};
  std::vector<unsigned long long> h(R);
  if (rg_launch<kBlock>(c->stream, (unsigned)R, k) || rg_copy_d2h(h.data(), rows, R * sizeof(unsigned long long), c->stream) || rg_stream_sync(c->stream))
    return RG_HIPFAIL(c, "state_checksum");
  unsigned long long sum = 0ull;
  for (size_t n = 0; n < R; ++n) sum += h[n];
  *out = sum;
  return RGPU_OK;
}

double rgpu_compute_dt(rgpu_ctx* c, int useU) {
  double v = 0;
  if (!c || rgpu_compute_inv_dt(c, useU, &v) != RGPU_OK) return std::numeric_limits<double>::quiet_NaN();
  return c->p.cfl / v;
}

int rgpu_step_pre(rgpu_ctx* c, int nStep, double dt, double totalTime) {
  (void)dt; (void)totalTime;
  RG_CHECK_CTX(c);
  if (!c->U[0]) return fail(c, RGPU_EINVAL, "context was not created");
  if (step_pre(c, nStep)) return RG_HIPFAIL(c, "step_pre");
  return RGPU_OK;
}
int rgpu_step_core(rgpu_ctx* c, int nStep, double dt, double totalTime) {
  RG_CHECK_CTX(c);
  if (!c->U[0]) return fail(c, RGPU_EINVAL, "context was not created");
  if (step_core(c, nStep, dt, totalTime)) return RG_HIPFAIL(c, "step_core");
  return RGPU_OK;
}
int rgpu_step_core_planes(rgpu_ctx* c, int nStep, double dt, double totalTime, int k_lo, int k_hi) {
  RG_CHECK_CTX(c);
  if (!c->U[0]) return fail(c, RGPU_EINVAL, "context was not created");
  if (step_core_planes(c, nStep, dt, totalTime, k_lo, k_hi)) return RG_HIPFAIL(c, "step_core_planes");
  return RGPU_OK;
}
int rgpu_step_core_planes_split(rgpu_ctx* c, int nStep, double dt, double totalTime, int k_lo, int k_hi, int what) {
  RG_CHECK_CTX(c);
  if (!c->U[0]) return fail(c, RGPU_EINVAL, "context was not created");
  if ((what & ~RGPU_CORE_SCAN) != RGPU_CORE_FLUXES && (what & ~RGPU_CORE_SCAN) != RGPU_CORE_UPDATE)
    return fail(c, RGPU_EINVAL, "step_core_planes_split: what must be RGPU_CORE_FLUXES or RGPU_CORE_UPDATE (| RGPU_CORE_SCAN)");
  if (step_core_planes(c, nStep, dt, totalTime, k_lo, k_hi, what)) return RG_HIPFAIL(c, "step_core_planes_split");
  return RGPU_OK;
}
int rgpu_step_core_planes_pair(rgpu_ctx* c, int nStep, double dt, double totalTime, int k_lo, int k_hi, int k_lo2, int k_hi2, int what) {
  RG_CHECK_CTX(c);
  if (!c->U[0]) return fail(c, RGPU_EINVAL, "context was not created");
  if ((what & ~RGPU_CORE_SCAN) != RGPU_CORE_FLUXES && (what & ~RGPU_CORE_SCAN) != RGPU_CORE_UPDATE)
    return fail(c, RGPU_EINVAL, "step_core_planes_pair: what must be RGPU_CORE_FLUXES or RGPU_CORE_UPDATE (| RGPU_CORE_SCAN)");
  if (k_hi > k_lo && k_hi2 > k_lo2 && k_lo2 < k_hi && k_lo < k_hi2) return fail(c, RGPU_EINVAL, "step_core_planes_pair: the two plane ranges overlap");
  if (step_core_planes(c, nStep, dt, totalTime, k_lo, k_hi, what, k_lo2, k_hi2)) return RG_HIPFAIL(c, "step_core_planes_pair");
  return RGPU_OK;
}
int rgpu_step_fill_planes_pair(rgpu_ctx* c, int nStep, double dt, double totalTime, int k_lo, int k_hi, int k_lo2, int k_hi2) {
  RG_CHECK_CTX(c);
  if (!c->U[0]) return fail(c, RGPU_EINVAL, "context was not created");
  if (!c->g.three_d) return fail(c, RGPU_EINVAL, "step_fill_planes: plane ranges need a 3D context");
  if (step_fill_planes(c, nStep, dt, totalTime, k_lo, k_hi, k_lo2, k_hi2)) return RG_HIPFAIL(c, "step_fill_planes_pair");
  return RGPU_OK;
}
int rgpu_step_dissipative(rgpu_ctx* c, int nStep, double dt, double totalTime) {
  RG_CHECK_CTX(c);
  if (!c->U[0]) return fail(c, RGPU_EINVAL, "context was not created");
  if (step_dissipative(c, nStep, dt, totalTime, false)) return RG_HIPFAIL(c, "step_dissipative");
  return RGPU_OK;
}
int rgpu_step_fill_planes(rgpu_ctx* c, int nStep, double dt, double totalTime, int k_lo, int k_hi) {
  RG_CHECK_CTX(c);
  if (!c->U[0]) return fail(c, RGPU_EINVAL, "context was not created");
  if (!c->g.three_d) return fail(c, RGPU_EINVAL, "step_fill_planes: plane ranges need a 3D context");
  if (k_lo < 0) k_lo = 0;
  if (k_hi > c->g.ksize) k_hi = c->g.ksize;
  if (k_hi <= k_lo) return RGPU_OK;
  if (step_fill_planes(c, nStep, dt, totalTime, k_lo, k_hi)) return RG_HIPFAIL(c, "step_fill_planes");
  return RGPU_OK;
}
int rgpu_step_post_a(rgpu_ctx* c, int nStep, double dt, double totalTime) {
  RG_CHECK_CTX(c);
  if (!c->U[0]) return fail(c, RGPU_EINVAL, "context was not created");
  if (step_post_a(c, nStep, dt, totalTime)) return RG_HIPFAIL(c, "step_post_a");
  return RGPU_OK;
}
int rgpu_step_post_b(rgpu_ctx* c, int nStep, double dt, double totalTime) {
  (void)dt; (void)totalTime;
  RG_CHECK_CTX(c);
  if (!c->U[0]) return fail(c, RGPU_EINVAL, "context was not created");
  if (step_post_b(c, nStep)) return RG_HIPFAIL(c, "step_post_b");
  return RGPU_OK;
}

int rgpu_godunov_unsplit(rgpu_ctx* c, int nStep, double dt, double totalTime) {
  RG_CHECK_CTX(c);
  if (!c->U[0]) return fail(c, RGPU_EINVAL, "context was not created");
  if (c->p.slab_count > 1) return fail(c, RGPU_EINVAL, "slab contexts must use rgpu_step_pre/core/post_a/post_b around the halo exchange");
  if (step_pre(c, nStep) || step_core(c, nStep, dt, totalTime) || step_dissipative(c, nStep, dt, totalTime) ||
      step_forcing(c, nStep, dt) || step_ou_forcing(c, (nStep + 1) % 2, dt) || step_post_a(c, nStep, dt, totalTime) || step_post_b(c, nStep))
    return RG_HIPFAIL(c, "godunov_unsplit");
  return RGPU_OK;
}

int rgpu_one_step_integration(rgpu_ctx* c, int* nStep, double* t, double* dt) {
  RG_CHECK_CTX(c);
  if (!nStep || !t || !dt) return fail(c, RGPU_EINVAL, "one_step_integration: null pointer");
  const double d = rgpu_compute_dt(c, *nStep % 2);
  if (!(d == d)) return RGPU_EHIP;
  *dt = d;
  const int rc = rgpu_godunov_unsplit(c, *nStep, d, *t);
  if (rc) return rc;
  *nStep += 1;
  *t += d;
  return RGPU_OK;
}

// ---- the device-side time step (csrc/step_clock_rec.h, hip/step_clock.h) ---------------------------------------------------
// Configuration: every kernel of the step that depends on dt or t reads the record, and nothing in the step needs the host between
// two steps.  2D: the fused step kernels (which also leave the ghost cells of their output: clock_ready).  3D: the z-marching
// sweeps, the MHD update, the shear remap and the fused ghost fill.  Not with gravity ((0.5 dt) g travels in DevParams), the
// dissipative stage, the forcings, the 2D rotating frame, the 2D jet, or the phase timers (they synchronise every launch anyway).
static bool clock_config_ok(rgpu_ctx* c) {
  const rgpu_params& p = c->p;
  if (c->timers_on || p.gravityEnabled != 0 || p.nu > 0 || (p.mhdEnabled && p.eta > 0) || p.randomForcingEnabled || p.ouForcingEnabled) return false;
  if (!rgpu_tiled::step_clock_supported()) return false;
  if (!c->g.three_d) return !c->g.rot && !p.enableJet;
  if (RG_SYNC_LAUNCH) return true;   // (host emulation: the record is resolved by value for every kernel)
  if (!p.mhdEnabled) return rgpu_tiled::hydro3d_sweep_covers(c->g);
  if (!rgpu_tiled::mhd3d_sweep_covers(c->g)) return false;
  FillXY f;
  return !(c->g.rot && c->g.shearbox) || fill_xy_plan(c, 0.0, 0.0, &f);   // the shearing ghost fill reads the record in its fused form only
}
// ... and the state U[parity]: its CFL maxima sit in the device slots; 2D: its ghost cells are the ones its kernel wrote
static bool clock_ready(rgpu_ctx* c, int parity) {
  if (c->p.slab_count != 1 || !clock_config_ok(c) || c->fused_dt_parity != parity) return false;
  if (!c->g.three_d) return c->fused_dt_slots == RG_DT_SLOTS && c->ghost_ok_parity == parity;
  return true;
}
static ClockConst clock_const(const rgpu_ctx* c) {
  const rgpu_params& p = c->p;
  ClockConst k;
  k.cfl = p.cfl;
  k.seed = 0.0;                                                                  // inv_dt_fetch: the floors of 1/dt
  if (p.mhdEnabled) k.seed = std::fmax(k.seed, p.smallc / std::fmin(p.dx, p.dy));
  if (p.enableJet) k.seed = std::fmax(k.seed, (p.ujet + p.cjet) / p.dx);
  k.dx = p.dx; k.dy = p.dy; k.dz = p.dz;
  k.Omega0 = p.Omega0; k.xlen = p.dx * p.nx; k.ylen = p.dy * p.ny;
  k.rot = c->g.rot; k.shear = (c->g.rot && c->g.shearbox && c->g.three_d) ? 1 : 0;
  return k;
}

int rgpu_device_time_step_ready(rgpu_ctx* c, int parity) { return (c && c->U[0] && clock_ready(c, parity & 1)) ? 1 : 0; }
int rgpu_clock_capable(rgpu_ctx* c) { return (c && c->U[0] && clock_config_ok(c)) ? 1 : 0; }

int rgpu_clock_open(rgpu_ctx* c, double t0, double tEnd) {
  RG_CHECK_CTX(c);
  if (!c->U[0]) return fail(c, RGPU_EINVAL, "context was not created");
  if (c->clk_n >= 0) return fail(c, RGPU_EINVAL, "clock_open: a batch is already open");
  if (!clock_config_ok(c)) return fail(c, RGPU_EUNSUPPORTED, "clock_open: this configuration takes its time step from the host");
  if (!c->d_clk) {
    if (rg_malloc((void**)&c->d_clk, rgpu_ctx::kClockBatch * sizeof(StepClock)) ||
        rg_host_alloc((void**)&c->h_clk, rgpu_ctx::kClockBatch * sizeof(StepClock))) return RG_HIPFAIL(c, "clock_open: records");
  }
  c->clk_n = 0; c->clk_t0 = t0; c->clk_tEnd = tEnd; c->clk_cur = 0;
  // The clock folded into the step kernel itself (ClockFold): the fused 2D HYDRO step on grids of at most two rounds of resident
  // workgroups.  Measured (profiles/r05_2d_clock_fold.txt): Kelvin-Helmholtz 512^2 (1369 workgroups) 0.0210 -> 0.0197 ms per step; but
  // every workgroup pays the fold (1024 slot reads, a barrier, the record) -- Orszag-Tang 512^2 (2145 workgroups of the MHD kernel)
  // 0.0439 -> 0.0458, 4096^2 +15 % -- and in the 3D MHD sweep (tried on the rotating path) the extra kernel argument alone moved the
  // register allocation of the z march: 25.1 -> 25.6 ms at 512^3.  Everything else keeps the one-workgroup clock kernel.
  {
    const int nwg = ((c->g.isize - 1 + 13) / 14) * ((c->g.jsize - 1 + 13) / 14);   // 16 x 16 thread tiles, 14 x 14 owned cells (tiled_hydro2d.h)
    c->fold_mode = !RG_SYNC_LAUNCH && rgpu_tiled::step_clock_fold_enabled() && !c->g.three_d && !c->p.mhdEnabled && nwg <= 2 * 768;
  }
  c->fold_pending = false;
  if (c->fold_mode) {   // the two slot arrays the first steps accumulate into / zero: clean (the host loop uses one array at a time)
    c->fold_phase0 = (int)((c->d_red - c->d_red_base) / RG_DT_SLOTS);
    for (int q = 1; q <= 2; ++q)
      if (rg_memset_async(c->d_red_base + ((c->fold_phase0 + q) % 3) * RG_DT_SLOTS, 0, RG_DT_SLOTS * sizeof(unsigned long long), c->stream)) { c->clk_n = -1; return RG_HIPFAIL(c, "clock_open"); }
  }
  return RGPU_OK;
}

int rgpu_clock_tick(rgpu_ctx* c) {
  RG_CHECK_CTX(c);
  if (c->clk_n < 0) return fail(c, RGPU_EINVAL, "clock_tick: no batch open");
  if (c->clk_n >= rgpu_ctx::kClockBatch) return fail(c, RGPU_EINVAL, "clock_tick: the batch is full");
  const int n = c->clk_n;
  if (c->fold_mode) {   // no launch: the step kernel that follows folds, forms and writes the record itself
    const int ph = (int)((c->d_red - c->d_red_base) / RG_DT_SLOTS);
    c->fold.prev = n ? c->d_clk + n - 1 : 0; c->fold.out = c->d_clk + n;
    // the step reads the maxima of its input from the current array and accumulates those of its output into the next one -- which
    // d_red names from here on (the step's update kernels, the slab driver's all-reduce before the next tick)
    c->fold.in = c->d_red; c->d_red = c->d_red_base + ((ph + 1) % 3) * RG_DT_SLOTS; c->fold.zero = c->d_red_base + ((ph + 2) % 3) * RG_DT_SLOTS;
    c->fold_pending = true;
    c->fold.k = clock_const(c); c->fold.t0 = c->clk_t0; c->fold.tEnd = c->clk_tEnd;
    c->clk_cur = c->d_clk + n;
    c->clk_n = n + 1;
    return RGPU_OK;
  }
  if (rgpu_tiled::launch_step_clock(c->stream, c->d_red, clock_const(c), c->clk_t0, c->clk_tEnd, n ? c->d_clk + n - 1 : 0, c->d_clk + n)) return RG_HIPFAIL(c, "clock_tick");
  c->clk_cur = c->d_clk + n;
  c->clk_n = n + 1;
  return RGPU_OK;
}

int rgpu_clock_stopped(rgpu_ctx* c) { return (c && stop_now(c)) ? 1 : 0; }

// host-checked: waits for the record of the last tick and returns its stop flag (0: the step runs; < 0: error)
int rgpu_clock_check(rgpu_ctx* c) {
  RG_CHECK_CTX(c);
  if (c->clk_n <= 0 || !c->clk_cur) return fail(c, RGPU_EINVAL, "clock_check: no tick in this batch");
  if (c->fold_mode) return 0;   // (the record is written by the step kernel that follows: nothing to read yet)
  if (RG_SYNC_LAUNCH) return c->clk_cur->stop;
  StepClock* h = c->h_clk + (c->clk_n - 1);
  if (rg_copy_d2h(h, c->d_clk + (c->clk_n - 1), sizeof(StepClock), c->stream) || rg_stream_sync(c->stream)) return RG_HIPFAIL(c, "clock_check");
  return h->stop;
}

int rgpu_clock_close(rgpu_ctx* c, int nStep0, int* ran, double* t, double* dt_last, double* dt_log, int* stop) {
  RG_CHECK_CTX(c);
  if (c->clk_n < 0) return fail(c, RGPU_EINVAL, "clock_close: no batch open");
  const int queued = c->clk_n;
  c->clk_n = -1; c->clk_cur = 0;
  const bool folded = c->fold_mode;
  c->fold_mode = false;
  if (ran) *ran = 0;
  if (stop) *stop = 0;
  if (queued > 0 && (rg_copy_d2h(c->h_clk, c->d_clk, (size_t)queued * sizeof(StepClock), c->stream) || rg_stream_sync(c->stream))) {
    state_modified(c);
    return RG_HIPFAIL(c, "clock_close: read-back of the records");
  }
  int r = 0;
  for (; r < queued && c->h_clk[r].stop == 0; ++r) {   // t accumulated in the order of the reference's loop
    if (dt_last) *dt_last = c->h_clk[r].dt;
    if (t) *t += c->h_clk[r].dt;
    if (dt_log) dt_log[r] = c->h_clk[r].dt;
  }
  if (ran) *ran = r;
  if (folded) c->d_red = c->d_red_base + ((c->fold_phase0 + r) % 3) * RG_DT_SLOTS;   // the array the last step that ran accumulated into
  if (r < queued) {
    // the steps behind a stop were no-ops (every kernel of a batch honours the flag, the stopping clock kernel left the slots alone):
    // the state of step nStep0 + r is the last one written, its CFL maxima are still in the slots, its ghost cells as its kernels left them
    if (stop) *stop = c->h_clk[r].stop;
    const int par = (nStep0 + r) % 2;
    c->scan_acc_parity = -1;
    c->fused_dt_parity = par;
    c->ghost_ok_parity = c->g.three_d ? -1 : par;
  }
  return RGPU_OK;
}

int rgpu_run_steps_log(rgpu_ctx* c, int nsteps, double tEnd, int* nStep, double* t, double* dt, double* dt_log) {
  RG_CHECK_CTX(c);
  if (!nStep || !t || !dt) return fail(c, RGPU_EINVAL, "run_steps: null pointer");
  int done = 0;
  while (done < nsteps && *t < tEnd) {
    const int parity = *nStep % 2;
    if (!clock_ready(c, parity)) {   // the reference's loop body (the first step of a run always comes through here)
      const int rc = rgpu_one_step_integration(c, nStep, t, dt);
      if (rc) return rc;
      if (dt_log) dt_log[done] = *dt;
      ++done;
      continue;
    }
    const int m = (nsteps - done < rgpu_ctx::kClockBatch) ? nsteps - done : (int)rgpu_ctx::kClockBatch;
    if (const int rc = rgpu_clock_open(c, *t, tEnd)) return rc;
    int queued = 0, rc = 0;
    const int n0 = *nStep;
    for (; queued < m; ++queued) {
      if ((rc = rgpu_clock_tick(c)) != 0) break;
      if (stop_now(c)) { ++queued; break; }   // (host emulation: the record is already there and says the loop has ended)
      // == rgpu_godunov_unsplit for this configuration, every dt / t dependence read from the record on the device
      const int n = n0 + queued;
      rc = (step_pre(c, n) || step_core(c, n, 0.0, 0.0) || step_post_a(c, n, 0.0, 0.0) || step_post_b(c, n)) ? RGPU_EHIP : 0;
      if (rc == 0 && c->fused_dt_parity != (n + 1) % 2) rc = RGPU_EHIP;   // (cannot happen: same configuration, same kernels)
      if (rc) { c->clk_n = queued; break; }   // the record of the step that failed to queue is not read back
    }
    // a launch that failed after `queued` complete steps were queued: those steps still run on the device -- read their records and
    // advance nStep / t / dt for them before reporting, so that the caller's step count and parity describe the device state
    const std::string launch_err = rc ? c->err + " " + rg_last_error_string() : std::string();
    int ran = 0, stop = 0;
    const int rc2 = rgpu_clock_close(c, n0, &ran, t, dt, dt_log ? dt_log + done : 0, &stop);
    if (rc2) return rc2;
    *nStep += ran;
    done += ran;
    if (rc) { state_modified(c); return fail(c, RGPU_EHIP, "run_steps: queueing a device-clock step: " + launch_err); }
    if (ran < queued) {
      if (stop >= 2) return fail(c, RGPU_EHIP, stop == 2 ? "run_steps: the time step is not a number" : "run_steps: 1/dt is not finite");
      break;
    }
  }
  return done;
}

int rgpu_run_steps(rgpu_ctx* c, int nsteps, double tEnd, int* nStep, double* t, double* dt) {
  return rgpu_run_steps_log(c, nsteps, tEnd, nStep, t, dt, 0);
}

int rgpu_synchronize(rgpu_ctx* c) {
  RG_CHECK_CTX(c);
  if (rg_stream_sync(c->stream)) return RG_HIPFAIL(c, "synchronize");
  return RGPU_OK;
}

int rgpu_enable_timers(rgpu_ctx* c, int enable) { RG_CHECK_CTX(c); c->timers_on = enable != 0; return RGPU_OK; }
int rgpu_get_timers(rgpu_ctx* c, double* secs, int n) {
  RG_CHECK_CTX(c);
  if (!secs) return RGPU_EINVAL;
  for (int i = 0; i < n && i < RGPU_T_COUNT; ++i) secs[i] = c->t_acc[i];
  return RGPU_OK;
}
int rgpu_reset_timers(rgpu_ctx* c) {
  RG_CHECK_CTX(c);
  for (int i = 0; i < RGPU_T_COUNT; ++i) { c->t_acc[i] = 0; c->t_calls[i] = 0; }
  return RGPU_OK;
}
const char* rgpu_timer_name(int which) {
  static const char* names[RGPU_T_COUNT] = {"boundaries", "prim", "elec", "trace", "flux", "emf", "update", "shear", "dt", "dissipative", "sweep"};
  return (which >= 0 && which < RGPU_T_COUNT) ? names[which] : "?";
}

int rgpu_dominant_kernel(rgpu_ctx* c, char* name, int name_len, double* avg_ms, long* launches) {
  RG_CHECK_CTX(c);
  int best = -1;
  for (int i = 0; i < RGPU_T_COUNT; ++i)
    if (c->t_calls[i] > 0 && (best < 0 || c->t_acc[i] > c->t_acc[best])) best = i;
  if (best < 0) return fail(c, RGPU_EINVAL, "no timed phase yet: call rgpu_enable_timers(ctx,1) and run steps");
  if (name && name_len > 0) std::snprintf(name, (size_t)name_len, "%s", rgpu_timer_name(best));
  if (avg_ms) *avg_ms = c->t_acc[best] * 1e3 / (double)c->t_calls[best];
  if (launches) *launches = c->t_calls[best];
  return RGPU_OK;
}

const char* rgpu_backend_name(void) { return RG_BACKEND_NAME; }
#ifdef RG_ARITH_FAST
const char* rgpu_arithmetic(void) { return "contracted"; }
#else
const char* rgpu_arithmetic(void) { return "exact"; }
#endif

int rgpu_selftest_arith(int n, const double* num, const double* den, double* quot, double* quot2, double* root, double* root2) {
  if (n <= 0 || !num || !den || !quot || !quot2 || !root || !root2) return RGPU_EINVAL;
  if (rg_device_count() < 1) return RGPU_ENODEVICE;
  double* d = 0;
  const size_t N = (size_t)n;
  if (rg_malloc((void**)&d, 6 * N * sizeof(double))) return RGPU_ENOMEM;
  const rg_stream_t s = (rg_stream_t)0;
  int rc = rg_copy_h2d(d, num, N * sizeof(double), s) || rg_copy_h2d(d + N, den, N * sizeof(double), s);
  K_selftest_arith k = {d, d + N, d + 2 * N, d + 3 * N, d + 4 * N, d + 5 * N};
  rc = rc || rg_launch<kBlock>(s, (unsigned)n, k);
  rc = rc || rg_copy_d2h(quot, d + 2 * N, N * sizeof(double), s) || rg_copy_d2h(quot2, d + 3 * N, N * sizeof(double), s) ||
       rg_copy_d2h(root, d + 4 * N, N * sizeof(double), s) || rg_copy_d2h(root2, d + 5 * N, N * sizeof(double), s) || rg_stream_sync(s);
  rg_free(d);
  return rc ? RGPU_EHIP : RGPU_OK;
}

int rgpu_selftest_alfven(const rgpu_params* p, int n, const double* states36, double* e_select, double* e_reference, int* route) {
  if (!p || n <= 0 || !states36 || !e_select || !e_reference || !route) return RGPU_EINVAL;
  if (rg_device_count() < 1) return RGPU_ENODEVICE;
  DevParams g;
  fill_dev_params(*p, &g);
  const size_t N = (size_t)n;
  double* d = 0; int* dr = 0;
  if (rg_malloc((void**)&d, 38 * N * sizeof(double)) || rg_malloc((void**)&dr, N * sizeof(int))) { rg_free(d); return RGPU_ENOMEM; }
  const rg_stream_t s = (rg_stream_t)0;
  K_selftest_alfven k = {g, d, d + 36 * N, d + 37 * N, dr, (unsigned)n};
  const int rc = rg_copy_h2d(d, states36, 36 * N * sizeof(double), s) || rg_launch<kBlock>(s, (unsigned)n, k) ||
                 rg_copy_d2h(e_select, d + 36 * N, N * sizeof(double), s) || rg_copy_d2h(e_reference, d + 37 * N, N * sizeof(double), s) ||
                 rg_copy_d2h(route, dr, N * sizeof(int), s) || rg_stream_sync(s);
  rg_free(d); rg_free(dr);
  return rc ? RGPU_EHIP : RGPU_OK;
}

int rgpu_set_option(const char* name, int value) {
  int* slot = rgpu::option_slot(name);
  if (!slot) return -1;
  const int old = *slot;
  *slot = value;
  return old;
}
int rgpu_get_option(const char* name) {
  const int* slot = rgpu::option_slot(name);
  return slot ? *slot : -1;
}

}  // extern "C"
