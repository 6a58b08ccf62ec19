// run_driver.h -- host-side run object mirroring the reference's run classes for the path in scope.
//
// class GodunovRun plays the role of HydroRunGodunov / MHDRunGodunov (HydroRunGodunov.h:91,177,
// MHDRunGodunov.h:120,253): init_simulation(), make_all_boundaries(), compute_dt(useU), godunov_unsplit(nStep,dt),
// oneStepIntegration(nStep,t,dt) -- the one pure virtual of HydroRunBase (HydroRunBase.h:433) -- and start().
// State lives on the device behind the C ABI (include/rgpu.h); the host mirror h_U is refreshed only for outputs
// (copyGpuToCpu, HydroRunBase.cpp:7217-7229).
#pragma once
#include <string>
#include <vector>

#include "../../../include/rgpu.h"
#include "host_params.h"
#include "ini_config.h"

namespace rgpu_host {
struct H5Box;

class GodunovRun {
 public:
  // slab_count > 1: slab slab_rank of a z-slab run; the stepping then goes through hooks (set_hooks) of the slab driver
  explicit GodunovRun(const IniConfig& cfg, int slab_rank = 0, int slab_count = 1);
  void set_hooks(const rgpuh_step_hooks& h) { hooks_ = h; hooked_ = true; }
  rgpu_ctx* ctx() { return ctx_; }
  ~GodunovRun();

  int init_simulation();                                   // initial condition -> h_U -> device U and U2
  void make_all_boundaries(int parity);
  double compute_dt(int useU);
  void godunov_unsplit(int nStep, double dt);
  void oneStepIntegration(int& nStep, double& t, double& dt);
  void copyGpuToCpu(int nStep);
  std::vector<double>& getDataHost() { return h_U_; }
  // time loop of start(); returns the number of steps; *mcell = "cell updates per second" / 1e6
  // attach (may be 0): called once the context holds its initial state, before the first ghost fill (rgpuh_run_hooked)
  int start(double* mcell_per_s, rgpuh_attach_fn attach = 0, void* user = 0);
  void outputVtk(int nStep);
  void outputVtkSlab(int nStep);   // z-slab runs: per-rank .vti + the .pvti index (HydroRunBaseMpi::outputVtk)
  // the reference's two raw single-variable formats: Xsmurf (density, doubles, current directory; HydroRunBase.cpp:2520-2562)
  // and NRRD (every variable as 32-bit floats, output directory; :4266-4335)
  void outputXsm(int nStep);
  void outputNrrd(int nStep);
  // restart=yes: read the interior fields, the step count and the time back from a .vti this driver wrote
  int inputVtk(const std::string& path);
  // raw dump <prefix>_NNNNNNN.rgr: what [output] outputHdf5=yes falls back to when no libhdf5 can be loaded; the role of the
  // reference's HDF5 files -- lossless state, optionally with the ghost cells ([output] ghostIncluded), step count and time
  void outputRestart(int nStep);
  // [output] outputHdf5=yes with a loadable libhdf5: the reference's file format (hdf5_io.h), interchangeable with its files
  void outputHdf5(int nStep);
  struct H5Box h5_box(int nx, int ny, int nz) const;
  int inputRestart(const std::string& path, bool* ghosts_read);
  int read_restart(const std::string& path, int nx, int ny, int nz, double* dst, bool* ghosts_read);
  void inputRestartUpscaled(const std::string& path, bool* ghosts_read);
  void save_forcing_process(int nStep);
  void restore_forcing_process(int nStep);
  void history(int nStep, double dt);                     // [history] enabled=yes: <outputDir>/<outputPrefix>_history.txt

  const rgpu_params& params() const { return p_; }
  double totalTime() const { return totalTime_; }

 private:
  IniConfig cfg_;
  rgpu_params p_;
  RunSettings rs_;
  rgpu_ctx* ctx_;
  std::vector<double> h_U_;
  double totalTime_;
  bool restart_has_ghosts_;
  bool warned_no_hdf5_, wrote_hdf5_;
  bool hooked_;
  rgpuh_step_hooks hooks_;
  bool slab() const { return p_.slab_count > 1; }
  void hook_check(int rc, const char* what);
  void agree_or_throw(const std::string& local_error, const char* what);
  void note_once(bool* flag, const char* msg);
  bool noted_vtk_, noted_hist_;
  void check(int rc, const char* what);
};

}  // namespace rgpu_host
