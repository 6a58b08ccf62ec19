// host_capi.cpp -- C entry points of the host side (parameter file -> rgpu_params, initial conditions).
// Declared in include/rgpu.h (rgpuh_*).
#include <cstdio>
#include <cstring>
#include <exception>
#include <string>

#include "../../../include/rgpu.h"
#include "host_params.h"
#include "ini_config.h"
#include "init_conditions.h"

namespace {
void set_err(char* err, int err_len, const std::string& msg) {
  if (err && err_len > 0) {
    std::snprintf(err, static_cast<size_t>(err_len), "%s", msg.c_str());
  }
}
int load(const char* ini_path, const char* overrides, rgpu_host::IniConfig* cfg, char* err, int err_len) {
  if (!ini_path) { set_err(err, err_len, "ini_path is NULL"); return RGPU_EINVAL; }
  const int rc = cfg->load_file(ini_path);
  if (rc < 0) { set_err(err, err_len, std::string("cannot open parameter file ") + ini_path); return RGPU_EINVAL; }
  if (rc > 0) { set_err(err, err_len, std::string("parse error in ") + ini_path + " at line " + std::to_string(rc)); return RGPU_EINVAL; }
  if (overrides) cfg->apply_overrides(overrides);
  return RGPU_OK;
}
}  // namespace

extern "C" {

size_t rgpu_state_elems(const rgpu_params* p) {
  if (!p) return 0;
  const size_t isize = p->nx + 2 * p->ghostWidth, jsize = p->ny + 2 * p->ghostWidth;
  const size_t ksize = (p->nz_global != 1) ? p->nz + 2 * p->ghostWidth : 1;
  return isize * jsize * ksize * static_cast<size_t>(p->nbVar);
}

int rgpuh_params_from_ini(const char* ini_path, const char* overrides, rgpu_params* out, char* err, int err_len) {
  if (!out) return RGPU_EINVAL;
  rgpu_host::IniConfig cfg;
  const int rc = load(ini_path, overrides, &cfg, err, err_len);
  if (rc) return rc;
  try {
    rgpu_host::RunSettings rs;
    const int rank = static_cast<int>(cfg.get_integer("slab", "rank", 0));
    const int count = static_cast<int>(cfg.get_integer("slab", "count", 1));
    rgpu_host::params_from_config(cfg, rank, count, out, &rs);
  } catch (const std::exception& e) {
    set_err(err, err_len, e.what());
    return RGPU_EUNSUPPORTED;
  }
  return RGPU_OK;
}

int rgpuh_run_settings(const char* ini_path, const char* overrides, int* nStepmax, double* tEnd, int* nOutput,
                       char* err, int err_len) {
  rgpu_host::IniConfig cfg;
  const int rc = load(ini_path, overrides, &cfg, err, err_len);
  if (rc) return rc;
  try {
    rgpu_host::RunSettings rs;
    rgpu_params p;
    rgpu_host::params_from_config(cfg, 0, 1, &p, &rs);
    if (nStepmax) *nStepmax = rs.nStepmax;
    if (tEnd) *tEnd = rs.tEnd;
    if (nOutput) *nOutput = rs.nOutput;
  } catch (const std::exception& e) {
    set_err(err, err_len, e.what());
    return RGPU_EUNSUPPORTED;
  }
  return RGPU_OK;
}

int rgpuh_init_condition(const char* ini_path, const char* overrides, const rgpu_params* p, double* hU, char* err,
                         int err_len) {
  if (!p || !hU) return RGPU_EINVAL;
  rgpu_host::IniConfig cfg;
  const int rc = load(ini_path, overrides, &cfg, err, err_len);
  if (rc) return rc;
  try {
    rgpu_host::init_condition(cfg, *p, hU);
  } catch (const std::exception& e) {
    set_err(err, err_len, e.what());
    return RGPU_EUNSUPPORTED;
  }
  return RGPU_OK;
}

int rgpuh_init_gravity(const char* ini_path, const char* overrides, const rgpu_params* p, double* hG, char* err, int err_len) {
  if (!p || !hG) return RGPU_EINVAL;
  rgpu_host::IniConfig cfg;
  const int rc = load(ini_path, overrides, &cfg, err, err_len);
  if (rc) return rc;
  try {
    return rgpu_host::init_gravity_field(cfg, *p, hG) ? 1 : 0;
  } catch (const std::exception& e) {
    set_err(err, err_len, e.what());
    return RGPU_EUNSUPPORTED;
  }
}

int rgpuh_init_forcing(const char* ini_path, const char* overrides, const rgpu_params* p, double* hF, char* err, int err_len) {
  if (!p || !hF) return RGPU_EINVAL;
  rgpu_host::IniConfig cfg;
  const int rc = load(ini_path, overrides, &cfg, err, err_len);
  if (rc) return rc;
  try {
    return rgpu_host::init_forcing_field(cfg, *p, hF) ? 1 : 0;
  } catch (const std::exception& e) {
    set_err(err, err_len, e.what());
    return RGPU_EUNSUPPORTED;
  }
}

}  // extern "C"
