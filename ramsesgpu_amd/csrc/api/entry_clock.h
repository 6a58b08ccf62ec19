// api/entry_clock.h -- entry points: the device-side time step (rgpu_clock_*, rgpu_run_steps).
#pragma once
extern "C" {
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
    // only for the library's own loop (rgpu_run_steps_log: the ghost cells of the input are known to be valid, no piece is queued
    // between the tick and the step kernel): with the record written by the step kernel, a piece queued in between by an external
    // driver would read the record of an earlier batch
    c->fold_mode = c->fold_request && !RG_SYNC_LAUNCH && rgpu_tiled::step_clock_fold_enabled() && !c->g.three_d && !c->p.mhdEnabled && nwg <= 2 * 768;
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
    c->fold_request = true;
    const int rc_open = rgpu_clock_open(c, *t, tEnd);
    c->fold_request = false;
    if (rc_open) return rc_open;
    int queued = 0, rc = 0;
    const int n0 = *nStep;
    for (; queued < m; ++queued) {
      if ((rc = rgpu_clock_tick(c)) != 0) break;
      if (stop_now(c)) { ++queued; break; }   // (host emulation: the record is already there and says the loop has ended)
      // == rgpu_godunov_unsplit for this configuration, every dt / t dependence read from the record on the device
      const int n = n0 + queued;
      if (!clock_ready(c, n % 2)) rc = RGPU_EHIP;   // (cannot happen: the step before left its CFL maxima and, in 2D, its ghost cells)
      if (rc == 0) rc = (step_pre(c, n) || step_core(c, n, 0.0, 0.0) || step_post_a(c, n, 0.0, 0.0) || step_post_b(c, n)) ? RGPU_EHIP : 0;
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

}  // extern "C"
