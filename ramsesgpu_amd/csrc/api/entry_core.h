// api/entry_core.h -- entry points of include/rgpu.h: context, transfers, ghost fill, CFL scan, history, the step and its pieces.
#pragma once
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

}  // extern "C"
