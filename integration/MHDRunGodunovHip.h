// MHDRunGodunovHip.h -- drop-in for MHDRunGodunov (src/hydro/MHDRunGodunov.h) on AMD MI355X: forwards the hot path
//   oneStepIntegration = compute_dt_mhd(nStep % 2) + godunov_unsplit(nStep, dt)      MHDRunGodunov.cpp:4077-4089
// and the ghost fill / host mirror calls around it to the C ABI of include/rgpu.h.  Everything else of the run class
// (ini parsing, initial conditions, outputs, history, the start() time loop) is inherited unchanged.
// Only VIRTUAL methods of the reference are overridden (HydroRunBase.h:405,425,433,512; MHDRunGodunov.h:246), so the
// reference sources need no edit: in src/euler_main.cpp:192  `new MHDRunGodunov(configMap)`  becomes
// `new MHDRunGodunovHip(configMap)`, built with -DUSE_DOUBLE (real_type.h:27-31) and linked with -lrgpu.
#ifndef MHD_RUN_GODUNOV_HIP_H_
#define MHD_RUN_GODUNOV_HIP_H_

#include "MHDRunGodunov.h"
#include "RgpuBinding.h"

namespace hydroSimu {

class MHDRunGodunovHip : public MHDRunGodunov {
 public:
  explicit MHDRunGodunovHip(ConfigMap& cfg) : MHDRunGodunov(cfg), ctx_(0) {
    rgpu_params p;
    // MHDRunGodunov::implementationVersion is private: re-read it the way the constructor does (MHDRunGodunov.cpp:159-181)
    int iv = configMap.getInteger("MHD", "implementationVersion", dimType == TWO_D ? 1 : 4);
    if (iv < 0 || iv > 4) iv = dimType == TWO_D ? 1 : 4;
    RgpuBinding::fill_params(*this, iv, 1, shearingBoxEnabled, &p);
    RgpuBinding::check(ctx_, rgpu_create(&p, &ctx_));
  }
  virtual ~MHDRunGodunovHip() { rgpu_destroy(ctx_); }

  // init_simulation fills h_U on the host exactly as today (MHDRunBase.cpp:1231-1364); then the device copy
  virtual int init_simulation(const std::string problemName) {
    const int step = MHDRunGodunov::init_simulation(problemName);
    RgpuBinding::check(ctx_, rgpu_upload(ctx_, h_U.data(), /*both=*/1));                         // d_U.copyFromHost(h_U), :1346-1351
    if (gravityEnabled) RgpuBinding::check(ctx_, rgpu_set_gravity_field(ctx_, h_gravity.data()));   // d_gravity.copyFromHost
    if (randomForcingEnabled) RgpuBinding::check(ctx_, rgpu_set_forcing_field(ctx_, h_randomForcing.data()));
    return step;
  }
  // start() fills the ghosts of the initial state once (MHDRunGodunov.cpp:3843-3851)
  virtual void make_all_boundaries(HostArray<real_t>&) {
    RgpuBinding::check(ctx_, rgpu_make_all_boundaries(ctx_, 0, totalTime, 0.0));
  }
  virtual void make_all_boundaries_shear(HostArray<real_t>&, real_t dt, int nStep) {
    RgpuBinding::check(ctx_, rgpu_make_all_boundaries(ctx_, nStep % 2, totalTime, dt));
  }
  // the hot path: dt from the device CFL scan, one unsplit step on the device (MHDRunGodunov.cpp:4077-4089)
  virtual void oneStepIntegration(int& nStep, real_t& t, real_t& dt) {
    RgpuBinding::check(ctx_, rgpu_one_step_integration(ctx_, &nStep, &t, &dt));
  }
  // outputs / history read the host mirror (HydroRunBase.cpp:7217-7229)
  virtual void copyGpuToCpu(int nStep = 0) {
    RgpuBinding::check(ctx_, rgpu_download(ctx_, (nStep % 2 == 0 ? h_U : h_U2).data(), nStep % 2));
  }

 private:
  rgpu_ctx* ctx_;
};

}  // namespace hydroSimu
#endif
