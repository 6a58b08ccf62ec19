// rgpu_comm.cpp -- the z-slab driver behind include/rgpu_comm.h: one process per GPU, each stepping its slab through the
// plane-ranged entry points of include/rgpu.h, ghost planes exchanged in place by the transport (rg_transport.h, chosen
// by include path: csrc/hip = RCCL; tests/emu = callbacks for the CPU tests).  Plain host C++: no kernels here.
//
// Schedule and its equivalence to the reference's fill order: see include/rgpu_comm.h and DESIGN.md section 6.
// (The Python class ramsesgpu_amd/slab.py is the same schedule over torch.distributed; it stays as the test harness.)
#include "../../../include/rgpu_comm.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "rg_transport.h"
#include "halo_ops.h"

using rgpu_transport::P2P;

struct rgpu_comm {
  rgpu_ctx* ctx;
  rgpu_params p;
  rgpu_transport::Comm* tc;
  int rank, nranks;
  int overlap;      // 0 serial, 1 overlapped (exchange behind the inner update), 2 boundary-first (exchange behind the inner sweep + update), -1: 1, or RGPU_COMM_SCHEDULE=1|2 from the environment
  int primed;     // parity of the state whose ghosts are all valid, -1 = none
  int scanned;    // parity of the state whose 1/dt sits in the context's device slot, -1 = none
  std::vector<P2P> ops[2];
  int scan_slots;   // > 0: the 1/dt maxima of the last step sit in that many device slots (fused scan), else in slot 0
  // != 0: a step piece failed on THIS rank (code kept here).  The rank keeps taking part in the exchanges (best effort, so
  // that no neighbour waits for planes that never come) and poisons the next 1/dt all-reduce with +inf: every rank then
  // returns an error from the same rgpu_comm_compute_dt instead of one rank leaving the others in a collective.
  bool fuse_scan;   // every rank's configuration lets the update pieces carry the CFL scan (agreed at create)
  bool clock_ok;    // every rank can keep the time step on the device between steps (agreed at create: a rank that batches while another
                    // takes the host loop would post collectives the other never matches)
  int poisoned;
  int exchanges_posted, exchanges_expected;   // halo exchanges of the current step: posted so far / what the neighbours will post
  long long clocked_steps;                    // steps whose time step came from the device record (rgpu_comm_run_steps)
  std::string err;
};

namespace {

int fail(rgpu_comm* cm, int code, const std::string& m) { if (cm) cm->err = m; return code; }
int ctx_fail(rgpu_comm* cm, int rc, const char* what) { return fail(cm, rc, std::string(what) + ": " + rgpu_last_error(cm->ctx)); }
int tr_fail(rgpu_comm* cm, const char* what) { return fail(cm, RGPU_EHIP, std::string(what) + ": " + cm->tc->err); }

#define RG_TRY(call, what) do { const int rc_ = (call); if (rc_) return ctx_fail(cm, rc_, what); } while (0)

bool rotating(const rgpu_comm* cm) { return cm->p.mhdEnabled && cm->p.Omega0 > 0; }
bool dissipative(const rgpu_comm* cm) { return cm->p.nu > 0 || (cm->p.mhdEnabled && cm->p.eta > 0); }

// send / receive descriptors of one state array (the arrays never move): per variable one contiguous chunk per face.
// Posting order = the order the peer's matching calls are posted in (grouped RCCL send / recv match in order per peer).
void build_ops(rgpu_comm* cm, int parity) {
  std::vector<P2P>& ops = cm->ops[parity];
  ops.clear();
  const rgpu_params& p = cm->p;
  const int gw = p.ghostWidth;
  const size_t plane = (size_t)(p.nx + 2 * gw) * (p.ny + 2 * gw);
  double* U = rgpu_device_state(cm->ctx, parity);
  std::vector<rgpu_transport::HaloOp> h;
  rgpu_transport::halo_ops(plane, gw, p.nz, p.nbVar, cm->rank, cm->nranks, p.bc[4] == RGPU_BC_COPY, p.bc[5] == RGPU_BC_COPY, h);
  for (size_t i = 0; i < h.size(); ++i) { const P2P o = {U + h[i].offset, h[i].count, h[i].peer, h[i].send}; ops.push_back(o); }
}

int exchange_start(rgpu_comm* cm, int parity) {
  const std::vector<P2P>& ops = cm->ops[parity & 1];
  cm->exchanges_posted += 1;
  if (ops.empty()) return 0;
  if (rgpu_transport::exchange_start(cm->tc, rgpu_stream_handle(cm->ctx), ops.data(), (int)ops.size())) return tr_fail(cm, "exchange_z_start");
  return 0;
}
int exchange_wait(rgpu_comm* cm) {
  if (rgpu_transport::exchange_wait(cm->tc, rgpu_stream_handle(cm->ctx))) return tr_fail(cm, "exchange_z_wait");
  return 0;
}
int exchange(rgpu_comm* cm, int parity) {
  const int rc = exchange_start(cm, parity);
  return rc ? rc : exchange_wait(cm);
}

int make_all_boundaries(rgpu_comm* cm, int parity, double t, double dt) {
  rgpu_ctx* c = cm->ctx;
  if (cm->p.shearingBoxEnabled) {
    RG_TRY(rgpu_make_boundaries(c, parity, RGPU_YDIR), "make_boundaries(Y)");
    RG_TRY(rgpu_make_boundaries_shear(c, parity, t, dt), "make_boundaries_shear");
    RG_TRY(rgpu_make_boundaries(c, parity, RGPU_ZDIR), "make_boundaries(Z)");   // physical z faces only
    if (int rc = exchange(cm, parity)) return rc;
    RG_TRY(rgpu_make_boundaries(c, parity, RGPU_YDIR), "make_boundaries(Y)");
  } else {
    RG_TRY(rgpu_make_boundaries(c, parity, RGPU_XDIR), "make_boundaries(X)");
    RG_TRY(rgpu_make_boundaries(c, parity, RGPU_YDIR), "make_boundaries(Y)");
    RG_TRY(rgpu_make_boundaries(c, parity, RGPU_ZDIR), "make_boundaries(Z)");
    if (int rc = exchange(cm, parity)) return rc;
  }
  cm->primed = parity;
  return 0;
}

// max over the slabs of the inverse time step: all-reduce of the device slot in place, ONE read-back
int compute_dt(rgpu_comm* cm, int useU, double* dt) {
  rgpu_ctx* c = cm->ctx;
  if (cm->scanned != useU && !cm->poisoned) {   // not accumulated plane range by plane range during the last step: full scan
    const int ks = cm->p.nz + 2 * cm->p.ghostWidth;
    const int rc = rgpu_inv_dt_accumulate(c, useU, 0, ks, 1);
    if (rc) { ctx_fail(cm, rc, "inv_dt_accumulate"); cm->poisoned = rc; }
    cm->scan_slots = 0;
  }
  cm->scanned = -1;
  cm->scan_slots = 0;
  if (cm->nranks > 1) {
    // ALWAYS all RGPU_DT_SLOTS slots (8 KB: latency-bound like 8 B): how many of them a rank's last step filled depends on its
    // state -- 1 after a full scan (first step, serial schedule), all of them after the fused scan of the overlapped schedule --
    // and a rank whose step piece failed has no state its peers share; a fixed count cannot mismatch.  Slot 0 is read in every
    // state, so +inf there reaches every rank.
    if (cm->poisoned) (void)rgpu_transport::poison_slot(cm->tc, rgpu_inv_dt_device_slot(c), rgpu_stream_handle(c));
    if (rgpu_transport::allreduce_max(cm->tc, rgpu_inv_dt_device_slot(c), RGPU_DT_SLOTS, rgpu_stream_handle(c))) return tr_fail(cm, "allreduce(1/dt)");
  }
  if (cm->poisoned) return cm->poisoned;   // message of the failed piece is in cm->err
  double inv = 0.0;
  RG_TRY(rgpu_inv_dt_result(c, &inv), "inv_dt_result");
  if (cm->nranks > 1 && !(inv < HUGE_VAL))
    return fail(cm, RGPU_EHIP, "1/dt is not finite after the all-reduce: another rank reported a failure (see its message), or the solution blew up");
  *dt = cm->p.cfl / inv;
  return 0;
}

int random_forcing(rgpu_comm* cm, int nStep, double dt) {
  const rgpu_params& p = cm->p;
  const int pout = (nStep + 1) % 2;
  double s[2];
  RG_TRY(rgpu_forcing_sums(cm->ctx, pout, s), "forcing_sums");
  if (cm->nranks > 1 && rgpu_transport::allreduce_sum_host(cm->tc, s, 2, rgpu_stream_handle(cm->ctx))) return tr_fail(cm, "allreduce(forcing sums)");
  double norm = 0.0;
  if (p.randomForcingEdot != 0) {
    const long long nb = (long long)p.nx * p.ny * p.nz_global;
    norm = (std::sqrt(s[0] * s[0] + s[1] * dt * p.randomForcingEdot * 2 * nb) - s[0]) / s[1];
  }
  RG_TRY(rgpu_add_forcing(cm->ctx, pout, norm), "add_forcing");
  return 0;
}

// exchange between the step pieces, nothing overlapped
int godunov_unsplit_serial(rgpu_comm* cm, int nStep, double dt, double t) {
  rgpu_ctx* c = cm->ctx;
  const bool rot = rotating(cm);
  cm->exchanges_expected = 1 + (dissipative(cm) ? 1 : 0);
  RG_TRY(rgpu_step_pre(c, nStep, dt, t), "step_pre");
  if (!rot) { if (int rc = exchange(cm, nStep % 2)) return rc; }          // plain path: ghosts of the INPUT
  RG_TRY(rgpu_step_core(c, nStep, dt, t), "step_core");
  if (dissipative(cm)) {
    // viscosity / resistivity work on the updated state with ALL its ghosts (mhd_godunov_unsplit_cpu_v3.cpp:662-668)
    if (int rc = make_all_boundaries(cm, (nStep + 1) % 2, t, dt)) return rc;
    RG_TRY(rgpu_step_dissipative(c, nStep, dt, t), "step_dissipative");
  }
  if (cm->p.randomForcingEnabled) { if (int rc = random_forcing(cm, nStep, dt)) return rc; }
  if (cm->p.ouForcingEnabled) RG_TRY(rgpu_step_ou_forcing(c, (nStep + 1) % 2, dt), "step_ou_forcing");   // same process on every rank
  RG_TRY(rgpu_step_post_a(c, nStep, dt, t), "step_post_a");
  if (rot) { if (int rc = exchange(cm, (nStep + 1) % 2)) return rc; }     // rotating path: ghosts of the OUTPUT
  RG_TRY(rgpu_step_post_b(c, nStep, dt, t), "step_post_b");
  cm->primed = rot ? (nStep + 1) % 2 : -1;
  cm->scanned = -1;
  return 0;
}

int godunov_unsplit_pieces(rgpu_comm* cm, int nStep, double dt, double t);

// A piece of the step that fails on this rank alone (a launch error) must not strand the neighbours in their ncclRecv: post
// the exchange of the output anyway (contents irrelevant: the run is over), remember the failure, report it through the next
// 1/dt all-reduce (compute_dt above).  exchanges_posted counts what the failed attempt had already posted.
int godunov_unsplit(rgpu_comm* cm, int nStep, double dt, double t) {
  if (cm->poisoned) return cm->poisoned;
  cm->exchanges_posted = 0;
  const int rc = godunov_unsplit_pieces(cm, nStep, dt, t);
  if (rc && cm->nranks > 1) {
    const std::string msg = cm->err;
    for (int n = cm->exchanges_posted; n < cm->exchanges_expected; ++n) (void)exchange(cm, (nStep + 1) % 2);
    cm->err = msg;
    cm->poisoned = rc;
  }
  return rc;
}

// The schedule of a step when the caller left the choice to the driver (rgpu_comm_set_overlap(-1), the default): RGPU_COMM_SCHEDULE=1|2
// if set, else by the thickness of the slab.  One-GPU probe with the link time emulated (profiles/r06_slab_probe.log, 512^2 planes, 103 MB
// per rank and step): slabs of 64 planes (N = 8 at 512^3) take 4.2-4.3 ms per step under schedule 2 whatever the link delivers, and
// 5.0 / 4.4 / 4.3 / 4.1 ms under schedule 1 at 40 / 60 / 80 / 100 GB/s per link -- the window of the inner update (~0.6 ms) is shorter
// than an exchange over xGMI at the rates RCCL's send / recv reach; slabs of 128 planes and more are ahead under schedule 1 at every
// rate (7.7 against 8.0 ms at 40 GB/s).  Every rank takes the same decision (all slabs have the same thickness).
int effective_schedule(const rgpu_comm* cm) {
  if (cm->overlap >= 0) return cm->overlap;
  static const char* e_mode = std::getenv("RGPU_COMM_SCHEDULE");   // 1 / 2 (include/rgpu_comm.h)
  static const int env_mode = e_mode ? std::atoi(e_mode) : -1;
  if (env_mode >= 1 && env_mode <= 2) return env_mode;
  return (cm->p.mhdEnabled && cm->p.nz <= 96) ? 2 : 1;
}

int godunov_unsplit_pieces(rgpu_comm* cm, int nStep, double dt, double t) {
  // the dissipative stage needs a second exchange inside the step, the random forcing a global sum and a change of the
  // whole updated state: both use the serial schedule
  if (!cm->overlap || dissipative(cm) || cm->p.randomForcingEnabled || cm->p.ouForcingEnabled) return godunov_unsplit_serial(cm, nStep, dt, t);
  rgpu_ctx* c = cm->ctx;
  const int pin = nStep % 2, pout = (nStep + 1) % 2;
  const bool rot = rotating(cm);
  cm->exchanges_expected = 1 + ((cm->primed != pin && !rot) ? 1 : 0);
  if (cm->primed != pin && !rot) {   // ghosts of the input not known to be valid (first step): fill them like the reference
    RG_TRY(rgpu_step_pre(c, nStep, dt, t), "step_pre");
    if (int rc = exchange(cm, pin)) return rc;
  }
  const int gw = cm->p.ghostWidth, nz = cm->p.nz, ks = nz + 2 * gw;
  // (boundary update ranges, planes to finish and send, inner update range) in array plane indices
  int bnd[2][2], snd[2][2], nb;
  bool has_inner;
  if (nz <= 2 * gw) { nb = 1; bnd[0][0] = 0; bnd[0][1] = ks; snd[0][0] = gw; snd[0][1] = nz + gw; has_inner = false; }
  else {
    // ghost planes behind a slab interface are not updated: the planes received in this step overwrite them whole (x / y ghosts
    // included), so their update -- old values + the CT of plane ksize - gw, what the reference leaves there until its next ghost
    // fill -- is 6 of the 70 planes of an N = 8 slab for nothing.  Physical z faces keep theirs (the reference's array contents).
    const bool lo_if = cm->p.bc[4] == RGPU_BC_COPY, hi_if = cm->p.bc[5] == RGPU_BC_COPY;
    nb = 2; bnd[0][0] = lo_if ? gw : 0; bnd[0][1] = 2 * gw; bnd[1][0] = nz; bnd[1][1] = hi_if ? nz + gw : ks;
    snd[0][0] = gw; snd[0][1] = 2 * gw; snd[1][0] = nz; snd[1][1] = nz + gw; has_inner = true;
  }
  const bool scan = !rot;   // rotating path: the reference's compute_dt sees the refilled ghosts -> full scan next step
  // fluxes / EMFs of the whole slab in one z-marching launch (3D MHD; a no-op for the solvers whose sweep is the whole step),
  // then the update range by range: the boundary-planes-first order costs no extra pipeline fill of the sweep
  // ... and the CFL scan of the new state rides in the update kernels (RGPU_CORE_SCAN) when the step allows it: no pass over
  // the output for the next compute_dt.  Otherwise, plain path: scan plane range by plane range before each fill.
  const int scan_flag = cm->fuse_scan ? RGPU_CORE_SCAN : 0;
  // Mode 2, boundary-first (3D MHD, the one solver whose update is a kernel of its own): the fluxes of the planes the boundary
  // updates read come from two short launches of the sweep, so that the exchange starts BEFORE the sweep of the inner planes and
  // hides behind it and the inner update (N = 8, 512^2 x 64 slab: a window of ~4 ms instead of the ~0.85 ms of the inner update
  // alone), for two extra pipeline fills of the z march (+0.3 ms at that size).  Mode -1: effective_schedule above.
  const int mode = effective_schedule(cm);
  const bool early = mode == 2 && has_inner && cm->p.mhdEnabled && nz > 4 * gw + 2;
  // both boundary ranges go through the *_pair entry points: one launch of the update kernel and one of the ghost fill for the two
  // of them -- what separates the end of the sweep from the start of the exchange is a handful of launches
  const int b2lo = nb == 2 ? bnd[1][0] : 0, b2hi = nb == 2 ? bnd[1][1] : 0, s2lo = nb == 2 ? snd[1][0] : 0, s2hi = nb == 2 ? snd[1][1] : 0;
  if (early) {
    // flux planes of a FLUXES call on [a, b) are [a, b + 1): low range -> planes gw .. 2 gw, high range -> nz .. nz + gw
    RG_TRY(rgpu_step_core_planes_pair(c, nStep, dt, t, bnd[0][0], bnd[0][1], b2lo, b2hi, RGPU_CORE_FLUXES | scan_flag), "step_core_planes(fluxes, boundary ranges)");
  } else {
    RG_TRY(rgpu_step_core_planes_split(c, nStep, dt, t, 0, ks, RGPU_CORE_FLUXES | scan_flag), "step_core_planes(fluxes)");
  }
  bool fused = cm->fuse_scan && rgpu_inv_dt_fused_active(c, pout) != 0;
  RG_TRY(rgpu_step_core_planes_pair(c, nStep, dt, t, bnd[0][0], bnd[0][1], b2lo, b2hi, RGPU_CORE_UPDATE | scan_flag), "step_core_planes(update, boundary ranges)");
  if (scan && !fused) for (int n = 0; n < nb; ++n) RG_TRY(rgpu_inv_dt_accumulate(c, pout, bnd[n][0], bnd[n][1], n == 0), "inv_dt_accumulate");
  RG_TRY(rgpu_step_fill_planes_pair(c, nStep, dt, t, snd[0][0], snd[0][1], s2lo, s2hi), "step_fill_planes(boundary ranges)");
  if (int rc = exchange_start(cm, pout)) return rc;
  if (has_inner) {
    // (boundary-first: the planes 2 gw + 1 .. nz - 1 are what the two short launches left)
    if (early) RG_TRY(rgpu_step_core_planes_split(c, nStep, dt, t, 2 * gw + 1, nz - 1, RGPU_CORE_FLUXES), "step_core_planes(fluxes, inner)");
    RG_TRY(rgpu_step_core_planes_split(c, nStep, dt, t, 2 * gw, nz, RGPU_CORE_UPDATE | scan_flag), "step_core_planes(update)");
    if (scan && !fused) RG_TRY(rgpu_inv_dt_accumulate(c, pout, 2 * gw, nz, 0), "inv_dt_accumulate");
    RG_TRY(rgpu_step_fill_planes(c, nStep, dt, t, 2 * gw, nz), "step_fill_planes");
  }
  if (int rc = exchange_wait(cm)) return rc;
  RG_TRY(rgpu_make_boundaries(c, pout, RGPU_ZDIR), "make_boundaries(Z)");   // physical z faces (+ 3D jet)
  cm->primed = pout;
  int nslots = 0;
  if (fused) {
    nslots = rgpu_inv_dt_fused_commit(c, pout);
    if (nslots <= 0) return fail(cm, RGPU_EHIP, "the accumulated CFL scan was lost between the update pieces");
  }
  cm->scanned = (fused || scan) ? pout : -1;
  cm->scan_slots = nslots;
  return 0;
}

}  // namespace

extern "C" {

int rgpu_comm_unique_id(char id[RGPU_COMM_ID_BYTES]) { return (id && rgpu_transport::unique_id(id) == 0) ? RGPU_OK : RGPU_EHIP; }

int rgpu_comm_create(rgpu_ctx* ctx, int rank, int nranks, const char id[RGPU_COMM_ID_BYTES], rgpu_comm** out) {
  if (!out) return RGPU_EINVAL;
  *out = 0;
  rgpu_comm* cm = new (std::nothrow) rgpu_comm();
  if (!cm) return RGPU_ENOMEM;
  *out = cm;   // returned on failure too, for rgpu_comm_last_error
  cm->ctx = ctx; cm->tc = 0; cm->rank = rank; cm->nranks = nranks; cm->overlap = -1; cm->primed = -1; cm->scanned = -1; cm->scan_slots = 0;
  cm->fuse_scan = false; cm->clock_ok = false; cm->poisoned = 0; cm->exchanges_posted = 0; cm->exchanges_expected = 0; cm->clocked_steps = 0;
  if (!ctx || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(cm, RGPU_EINVAL, "comm_create: bad arguments");
  if (rgpu_get_params(ctx, &cm->p)) return fail(cm, RGPU_EINVAL, "comm_create: no parameters in the context");
  if (cm->p.nz_global == 1) return fail(cm, RGPU_EUNSUPPORTED, "2D problems do not shard: run replicas");
  if (cm->p.slab_rank != rank || cm->p.slab_count != nranks) return fail(cm, RGPU_EINVAL, "comm_create: the context was created for another slab_rank / slab_count");
  (void)rgpu_stream_handle(ctx);   // makes the context's device current (every rgpu entry point does)
  if (rgpu_transport::create(&cm->tc, rank, nranks, id)) return fail(cm, RGPU_EHIP, "transport: " + (cm->tc ? cm->tc->err : std::string("allocation")));
  build_ops(cm, 0);
  build_ops(cm, 1);
  // the fused CFL scan changes how many device slots the 1/dt all-reduce carries: all ranks or none (an end slab with an open
  // / stratified z face on the rotating path cannot fuse it, the inner slabs could)
  // ... and so does the packed exchange (one message per peer instead of one per chunk): its staging buffers are sized and
  // allocated HERE, not inside the first step, and a rank that cannot have them takes every rank to the in-place exchange
  // ... and so does the device-side time step (a batch posts its collectives without a host turn: every rank batches, or none)
  double cannot[3] = {rgpu_inv_dt_fusable(ctx) ? 0.0 : 1.0, 0.0, (rgpu_clock_capable(ctx) && rgpu_get_option("step_clock") != 0) ? 0.0 : 1.0};
  for (int par = 0; par < 2; ++par)
    if (rgpu_transport::prepare_exchange(cm->tc, cm->ops[par].data(), (int)cm->ops[par].size())) cannot[1] = 1.0;
  if (nranks > 1 && rgpu_transport::allreduce_sum_host(cm->tc, cannot, 3, rgpu_stream_handle(ctx))) return tr_fail(cm, "comm_create: allreduce");
  cm->fuse_scan = cannot[0] < 0.5;
  if (cannot[1] > 0.5) rgpu_transport::disable_pack(cm->tc);
  cm->clock_ok = cannot[2] < 0.5;
  return RGPU_OK;
}

void rgpu_comm_destroy(rgpu_comm* cm) {
  if (!cm) return;
  // a rank whose step failed may hold a collective its peers never match (see `poisoned`): abort, do not wait
  if (cm->tc && cm->poisoned) rgpu_transport::abort_comm(cm->tc);
  if (cm->tc) rgpu_transport::destroy(cm->tc);
  delete cm;
}

const char* rgpu_comm_last_error(rgpu_comm* cm) { return cm ? cm->err.c_str() : "null communicator"; }
const char* rgpu_comm_transport_name(void) { return RG_TRANSPORT_NAME; }
int rgpu_comm_set_device(int device) { rgpu_transport::set_device(device); return RGPU_OK; }
int rgpu_comm_info(rgpu_comm* cm, int* transport_ranks, int* transport_rank, int* device, char* pci_bus_id, int pci_len) {
  if (!cm || !cm->tc) return RGPU_EINVAL;
  if (rgpu_transport::info(cm->tc, transport_ranks, transport_rank, device, pci_bus_id, pci_len)) return tr_fail(cm, "comm_info");
  return RGPU_OK;
}

#define RG_CHECK_CM(cm) do { if (!(cm) || !(cm)->tc) return RGPU_EINVAL; } while (0)

int rgpu_comm_exchange_z_start(rgpu_comm* cm, int parity) { RG_CHECK_CM(cm); return exchange_start(cm, parity); }
int rgpu_comm_exchange_z_wait(rgpu_comm* cm) { RG_CHECK_CM(cm); return exchange_wait(cm); }
int rgpu_comm_make_all_boundaries(rgpu_comm* cm, int parity, double totalTime, double dt) { RG_CHECK_CM(cm); return make_all_boundaries(cm, parity & 1, totalTime, dt); }
int rgpu_comm_compute_dt(rgpu_comm* cm, int useU, double* dt) { RG_CHECK_CM(cm); if (!dt) return RGPU_EINVAL; return compute_dt(cm, useU & 1, dt); }
int rgpu_comm_godunov_unsplit(rgpu_comm* cm, int nStep, double dt, double totalTime) { RG_CHECK_CM(cm); return godunov_unsplit(cm, nStep, dt, totalTime); }
int rgpu_comm_rccl_version(rgpu_comm* cm) { return (cm && cm->tc) ? rgpu_transport::version(cm->tc) : 0; }
double rgpu_comm_last_exchange_ms(rgpu_comm* cm) { return (cm && cm->tc && !cm->ops[0].empty()) ? rgpu_transport::last_exchange_ms(cm->tc) : -1.0; }
long long rgpu_comm_halo_bytes(rgpu_comm* cm) {
  if (!cm || !cm->tc) return 0;
  long long b = 0;
  for (size_t i = 0; i < cm->ops[0].size(); ++i) if (cm->ops[0][i].send) b += (long long)(cm->ops[0][i].count * sizeof(double));
  return b;
}
long long rgpu_comm_clocked_steps(rgpu_comm* cm) { return cm ? cm->clocked_steps : 0; }
int rgpu_comm_schedule(rgpu_comm* cm) { return cm ? effective_schedule(cm) : -1; }
int rgpu_comm_set_overlap(rgpu_comm* cm, int overlap) { RG_CHECK_CM(cm); cm->overlap = (overlap < -1 || overlap > 2) ? 1 : overlap; return RGPU_OK; }

int rgpu_comm_one_step_integration(rgpu_comm* cm, int* nStep, double* t, double* dt) {
  RG_CHECK_CM(cm);
  if (!nStep || !t || !dt) return fail(cm, RGPU_EINVAL, "one_step_integration: null pointer");
  double d = 0.0;
  if (int rc = compute_dt(cm, *nStep % 2, &d)) return rc;
  *dt = d;
  if (int rc = godunov_unsplit(cm, *nStep, d, *t)) {
    // tell the other ranks now (their next collective is the 1/dt all-reduce of the next step): see `poisoned`
    if (cm->nranks > 1 && cm->poisoned) { double dummy; (void)compute_dt(cm, (*nStep + 1) % 2, &dummy); }
    return rc;
  }
  *nStep += 1;
  *t += d;
  return RGPU_OK;
}

// The reference's loop body for a run of steps, the time step on the device (rgpu.h: rgpu_clock_*): per step
//   ncclAllReduce(max) of the RGPU_DT_SLOTS 1/dt slots in place  ->  clock kernel (fold, dt, t += dt, the "t < tEnd" test, record)
//   ->  the step pieces of the schedule (they read the record on the device)  ->  halo exchange on the side stream
// all queued on the streams without a host turn; the host reads the records of the batch once.  A step qualifies when the CFL maxima
// of its input sit in the slots (the update pieces of the previous step carried the scan: schedules 1 and 2, all ranks fusable) and the
// context's configuration lets its kernels read the record (rgpu_clock_capable); every other step -- the first of a run, the
// serial schedule -- is the plain rgpu_comm_one_step_integration.  Every rank takes the same decisions (configuration and step
// count only), so the collectives pair up.
int rgpu_comm_run_steps(rgpu_comm* cm, int nsteps, double tEnd, int* nStep, double* t, double* dt, double* dt_log) {
  RG_CHECK_CM(cm);
  if (!nStep || !t || !dt) return fail(cm, RGPU_EINVAL, "run_steps: null pointer");
  rgpu_ctx* c = cm->ctx;
  int done = 0;
  while (done < nsteps && *t < tEnd) {
    const int useU = *nStep % 2;
    // (clock_ok: agreed between the ranks at create; a context whose phase timers were switched on since then -- rgpu_clock_capable --
    // must have had them switched on on every rank, as bench.py and the run driver do)
    const bool batch = cm->clock_ok && !cm->poisoned && cm->scanned == useU && cm->scan_slots > 0 && cm->fuse_scan && cm->overlap != 0 &&
                       !dissipative(cm) && !cm->p.randomForcingEnabled && !cm->p.ouForcingEnabled && rgpu_clock_capable(c);
    if (!batch) {
      if (int rc = rgpu_comm_one_step_integration(cm, nStep, t, dt)) return rc;
      if (dt_log) dt_log[done] = *dt;
      ++done;
      continue;
    }
    const int m = (nsteps - done < RGPU_CLOCK_BATCH) ? nsteps - done : RGPU_CLOCK_BATCH;
    RG_TRY(rgpu_clock_open(c, *t, tEnd), "clock_open");
    const int n0 = *nStep;
    int queued = 0, rc = 0, first_fail = -1;
    bool told = false;   // this rank failed AND a poisoned all-reduce has gone out since
    for (; queued < m; ++queued) {
      const int n = n0 + queued;
      if (rc != 0) {
        // a piece failed on THIS rank at an earlier step of the batch: the other ranks keep queueing theirs, so keep pairing up with
        // them -- +inf into the 1/dt all-reduce (their records say stop = 3 from here on) and the one halo exchange each of their
        // no-op steps still posts.  A synchronous backend is different: there the others SEE the stop at their tick and leave
        // before that exchange; this rank ticks too (its record says 3 as well) and leaves at the same place.
        const std::string msg = cm->err;
        (void)rgpu_transport::poison_slot(cm->tc, rgpu_inv_dt_device_slot(c), rgpu_stream_handle(c));
        const bool dead = rgpu_transport::allreduce_max(cm->tc, rgpu_inv_dt_device_slot(c), RGPU_DT_SLOTS, rgpu_stream_handle(c)) != 0;
        told = !dead;
        bool left = false;
        if (!dead) {
          (void)rgpu_clock_tick(c);
          left = rgpu_clock_stopped(c) != 0;
          if (!left) (void)exchange(cm, (n + 1) % 2);
        }
        cm->err = msg;
        if (dead || left) break;
        continue;
      }
      if (cm->nranks > 1 && rgpu_transport::allreduce_max(cm->tc, rgpu_inv_dt_device_slot(c), RGPU_DT_SLOTS, rgpu_stream_handle(c))) { rc = tr_fail(cm, "allreduce(1/dt)"); first_fail = queued; break; }
      if ((rc = rgpu_clock_tick(c)) != 0) {
        ctx_fail(cm, rc, "clock_tick");
        const std::string msg = cm->err;
        if (cm->nranks > 1) (void)exchange(cm, (n + 1) % 2);      // what the neighbours' step n posts
        cm->err = msg;
      } else if (rgpu_clock_stopped(c) || (queued == 0 && cm->nranks > 1 && rgpu_clock_check(c) != 0)) {
        // the record says the loop has ended -- on every rank alike (the 1/dt it was formed from is all-reduced).  Synchronous
        // backends know at once; on the device the FIRST step of every batch is checked on the host (one synchronisation per
        // RGPU_CLOCK_BATCH steps): a rank whose previous, unbatched step failed has posted one poisoned all-reduce and left -- without
        // this check its peers would queue a whole batch of no-op steps whose exchanges and all-reduces nobody matches, and hang
        ++queued;
        break;
      } else {
        cm->scanned = -1; cm->scan_slots = 0;
        rc = godunov_unsplit(cm, n, 0.0, 0.0);      // the pieces read dt and t from the record; on failure it has posted the step's exchanges
        if (rc == 0 && !(cm->scanned == (n + 1) % 2 && cm->scan_slots > 0)) rc = fail(cm, RGPU_EHIP, "run_steps: the update pieces did not carry the CFL scan");
      }
      if (rc) {
        first_fail = queued;
        cm->poisoned = rc;
        if (cm->nranks <= 1) break;
      }
    }
    int ran = 0, stop = 0;
    const std::string msg = cm->err;
    const int rc2 = rgpu_clock_close(c, n0, &ran, t, dt, dt_log ? dt_log + done : 0, &stop);
    if (rc2) return ctx_fail(cm, rc2, "clock_close");
    if (rc) {   // this rank's own failure: the steps before it ran; the run is over on every rank (poisoned)
      // A failure in the LAST step of the batch has not told anybody yet: the others finished their batch in good health, and
      // unless the records (theirs too, up to the failed step) say that the run has reached tEnd, their next collective is the 1/dt
      // all-reduce of the next turn -- of this call or the caller's next one, batched or not.  Pair it, poisoned, as
      // rgpu_comm_one_step_integration does: the first record of a batch is host-checked, an unbatched turn reads 1/dt on the host.
      // (Should the others never take that turn -- nstepmax reached -- the all-reduce stays unmatched on THIS rank only, whose run
      // is over; rgpu_comm_destroy aborts a poisoned communicator instead of waiting for it.)
      if (cm->nranks > 1 && !told && stop == 0 && *t < tEnd) {
        (void)rgpu_transport::poison_slot(cm->tc, rgpu_inv_dt_device_slot(c), rgpu_stream_handle(c));
        (void)rgpu_transport::allreduce_max(cm->tc, rgpu_inv_dt_device_slot(c), RGPU_DT_SLOTS, rgpu_stream_handle(c));
      }
      cm->err = msg;
      *nStep += first_fail < ran ? first_fail : ran;
      cm->primed = -1; cm->scanned = -1; cm->scan_slots = 0;
      return rc;
    }
    *nStep += ran;
    done += ran;
    cm->clocked_steps += ran;
    if (ran < queued) {
      // steps behind a stop were no-ops: the state of step n0 + ran is the last one written, with the ghosts and the (all-reduced) CFL
      // maxima its step left -- rgpu_clock_close has told the context; the exchanges the no-op steps posted carried unchanged planes
      const int par = (n0 + ran) % 2;
      cm->primed = par; cm->scanned = par; cm->scan_slots = RGPU_DT_SLOTS;
      if (stop == 3) { cm->poisoned = RGPU_EHIP; return fail(cm, RGPU_EHIP, "1/dt is not finite after the all-reduce: another rank reported a failure (see its message), or the solution blew up"); }
      if (stop == 2) return fail(cm, RGPU_EHIP, "run_steps: the time step is not a number");
      break;
    }
  }
  return done;
}

int rgpu_comm_history_mri(rgpu_comm* cm, int parity, double* out) {
  RG_CHECK_CM(cm);
  if (!out) return fail(cm, RGPU_EINVAL, "history_mri: null pointer");
  const rgpu_params& p = cm->p;
  if (!p.mhdEnabled) return fail(cm, RGPU_EUNSUPPORTED, "history diagnostics are defined for MHD runs");
  const int gw = p.ghostWidth, is = p.nx + 2 * gw, NQ = 9;
  std::vector<double> cols((size_t)NQ * is), rcol(is), mvx(is), mvy(is);
  RG_TRY(rgpu_history_columns(cm->ctx, parity, cols.data()), "history_columns");
  void* s = rgpu_stream_handle(cm->ctx);
  if (cm->nranks > 1 && rgpu_transport::allreduce_sum_host(cm->tc, cols.data(), NQ * is, s)) return tr_fail(cm, "allreduce(history columns)");
  const bool three_d = p.nz_global != 1;
  const double dTau = three_d ? p.dx * p.dy * p.dz / (p.xMax - p.xMin) / (p.yMax - p.yMin) / (p.zMax - p.zMin)
                              : p.dx * p.dy / (p.xMax - p.xMin) / (p.yMax - p.yMin);
  const double nyz = (double)p.ny * (three_d ? p.nz_global : 1);
  for (int i = 0; i < is; ++i) { mvx[i] = cols[(size_t)1 * is + i] / nyz; mvy[i] = cols[(size_t)2 * is + i] / nyz; }
  RG_TRY(rgpu_history_reynolds(cm->ctx, parity, mvx.data(), mvy.data(), dTau, rcol.data()), "history_reynolds");
  if (cm->nranks > 1 && rgpu_transport::allreduce_sum_host(cm->tc, rcol.data(), is, s)) return tr_fail(cm, "allreduce(history reynolds)");
  double sum[9], reyn = 0.0;
  for (int q = 0; q < NQ; ++q) { sum[q] = 0.0; for (int i = gw; i < is - gw; ++i) sum[q] += cols[(size_t)q * is + i]; }
  for (int i = gw; i < is - gw; ++i) reyn += rcol[i];
  out[0] = sum[0] * dTau; out[1] = sum[4] * dTau; out[2] = reyn; out[3] = sum[3] * dTau / 2.;
  out[4] = sum[5] * dTau; out[5] = sum[6] * dTau; out[6] = sum[7] * dTau; out[7] = sum[8];
  return RGPU_OK;
}

int rgpu_comm_history_turbulence(rgpu_comm* cm, int parity, double* out) {
  RG_CHECK_CM(cm);
  if (!out) return fail(cm, RGPU_EINVAL, "history_turbulence: null pointer");
  const rgpu_params& p = cm->p;
  double s[18];
  RG_TRY(rgpu_history_turbulence_sums(cm->ctx, parity, s), "history_turbulence_sums");
  const double dTau = p.dx * p.dy * p.dz / (p.xMax - p.xMin) / (p.yMax - p.yMin) / (p.zMax - p.zMin);   // zMax - zMin: the whole box
  // this rank's values, scaled like the reference does before its MPI_Reduce calls
  const double mass = s[0] * dTau, eKin = s[1] * dTau, mean_v2 = s[2] * dTau, eMag = s[3] * dTau, helicity = s[4] * dTau, divB = s[17];
  const double mB[3] = {s[5] * dTau, s[6] * dTau, s[7] * dTau}, mrv[3] = {s[8] * dTau, s[9] * dTau, s[10] * dTau};
  const double mB_norm = std::sqrt(mB[0] * mB[0] + mB[1] * mB[1] + mB[2] * mB[2]);
  double t[8] = {mass, mean_v2, eKin, eMag, mB_norm, mB[0], mB[1], mB[2]};
  if (cm->nranks > 1 && rgpu_transport::allreduce_sum_host(cm->tc, t, 8, rgpu_stream_handle(cm->ctx))) return tr_fail(cm, "allreduce(history turbulence)");
  const double pi = 2 * std::asin(1.0);
  out[0] = t[0]; out[1] = divB; out[2] = t[2]; out[3] = t[3]; out[4] = helicity; out[5] = t[4];
  out[6] = t[5]; out[7] = t[6]; out[8] = t[7]; out[9] = mrv[0]; out[10] = mrv[1]; out[11] = mrv[2];
  out[12] = std::sqrt(t[1]) / p.cIso;                                   // Ma_s
  out[13] = std::sqrt(t[1]) / (t[4] / std::sqrt(4 * pi * t[0]));        // Ma_alfven
  return RGPU_OK;
}

// euler_hip --slabs: the run loop of the single-GPU front end (rgpuh_run_hooked in librgpu: initial condition or restart of this
// slab, the reference's time loop, HDF5 outputs of the whole box written slab after slab) stepping through this driver
namespace {
struct SlabAttach {
  int rank, nranks;
  const char* id;
  rgpu_comm* cm;
  std::string err;
};
int hook_make_all_boundaries(void* self, int parity, double t, double dt) { return rgpu_comm_make_all_boundaries(static_cast<SlabAttach*>(self)->cm, parity, t, dt); }
int hook_compute_dt(void* self, int useU, double* dt) { return rgpu_comm_compute_dt(static_cast<SlabAttach*>(self)->cm, useU, dt); }
int hook_one_step(void* self, int* nStep, double* t, double* dt) { return rgpu_comm_one_step_integration(static_cast<SlabAttach*>(self)->cm, nStep, t, dt); }
int hook_run_steps(void* self, int nsteps, double tEnd, int* nStep, double* t, double* dt) { return rgpu_comm_run_steps(static_cast<SlabAttach*>(self)->cm, nsteps, tEnd, nStep, t, dt, 0); }
int hook_history_mri(void* self, int parity, double* out) { return rgpu_comm_history_mri(static_cast<SlabAttach*>(self)->cm, parity, out); }
int hook_history_turbulence(void* self, int parity, double* out) { return rgpu_comm_history_turbulence(static_cast<SlabAttach*>(self)->cm, parity, out); }
int hook_barrier(void* self) {
  rgpu_comm* cm = static_cast<SlabAttach*>(self)->cm;
  if (rgpu_synchronize(cm->ctx)) return RGPU_EHIP;
  return rgpu_transport::barrier(cm->tc, rgpu_stream_handle(cm->ctx)) ? RGPU_EHIP : 0;
}
// number of ranks with local_failed != 0: a SUM all-reduce of the flags (host values through the transport's scratch)
int hook_agree(void* self, int local_failed) {
  rgpu_comm* cm = static_cast<SlabAttach*>(self)->cm;
  if (rgpu_synchronize(cm->ctx)) return RGPU_EHIP;
  double f = local_failed ? 1.0 : 0.0;
  if (cm->nranks > 1 && rgpu_transport::allreduce_sum_host(cm->tc, &f, 1, rgpu_stream_handle(cm->ctx))) return RGPU_EHIP;
  return (int)(f + 0.5);
}
const char* hook_last_error(void* self) {
  SlabAttach* a = static_cast<SlabAttach*>(self);
  if (!a->cm) return a->err.c_str();
  return !a->cm->err.empty() ? a->cm->err.c_str() : rgpu_last_error(a->cm->ctx);
}
int slab_attach(void* user, rgpu_ctx* ctx, rgpuh_step_hooks* h) {
  SlabAttach* a = static_cast<SlabAttach*>(user);
  h->self = a;
  h->last_error = hook_last_error;
  const int rc = rgpu_comm_create(ctx, a->rank, a->nranks, a->id, &a->cm);
  if (rc) { a->err = a->cm ? a->cm->err : "rgpu_comm_create failed"; if (a->cm) { rgpu_comm_destroy(a->cm); a->cm = 0; } return rc; }
  {   // one rank <-> one device (HydroMpiParameters.cpp:196-201): say what the transport itself reports
    int n = 0, r = -1, d = -1;
    char pci[64] = {0};
    if (rgpu_comm_info(a->cm, &n, &r, &d, pci, (int)sizeof(pci)) == RGPU_OK) {
      std::printf("rank %d/%d -> HIP device %d (PCI %s), %s communicator of %d rank%s\n", r, a->nranks, d, pci[0] ? pci : "?",
                  std::strcmp(RG_TRANSPORT_NAME, "rccl") ? RG_TRANSPORT_NAME : "RCCL", n, n == 1 ? "" : "s");
      std::fflush(stdout);
      if (n != a->nranks || r != a->rank) { a->err = "the transport reports another communicator size / rank than the launch asked for"; rgpu_comm_destroy(a->cm); a->cm = 0; return RGPU_EINVAL; }
    }
  }
  h->make_all_boundaries = hook_make_all_boundaries;
  h->compute_dt = hook_compute_dt;
  h->one_step_integration = hook_one_step;
  h->run_steps = hook_run_steps;
  h->barrier = hook_barrier;
  h->agree = hook_agree;
  h->history_mri = hook_history_mri;
  h->history_turbulence = hook_history_turbulence;
  return 0;
}
void slab_detach(void* user) {
  SlabAttach* a = static_cast<SlabAttach*>(user);
  if (a->cm) { rgpu_comm_destroy(a->cm); a->cm = 0; }
}
}  // namespace

int rgpuh_run_slabs(const char* ini_path, const char* overrides, int rank, int nranks, int device,
                    const char id[RGPU_COMM_ID_BYTES], double* mcell_per_s, char* err, int err_len) {
  if (!ini_path || !id) { if (err && err_len > 0) std::snprintf(err, (size_t)err_len, "run_slabs: null argument"); return RGPU_EINVAL; }
  rgpu_transport::set_device(device);
  SlabAttach a = {rank, nranks, id, 0, std::string()};
  return rgpuh_run_hooked(ini_path, overrides, rank, nranks, slab_attach, slab_detach, &a, mcell_per_s, err, err_len);
}

}  // extern "C"
