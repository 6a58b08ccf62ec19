// api/history.h -- history diagnostics (row / column sums on the device) and the forcings.  See api/ctx.h.
#pragma once
namespace {
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
