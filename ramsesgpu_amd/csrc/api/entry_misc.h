// api/entry_misc.h -- entry points: synchronisation, phase timers, names, the self-tests of the device arithmetic, options.
#pragma once
extern "C" {
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
