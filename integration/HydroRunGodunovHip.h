// HydroRunGodunovHip.h -- drop-in for HydroRunGodunov (src/hydro/HydroRunGodunov.h) on AMD MI355X: same binding as
// MHDRunGodunovHip.h for the hydro step   oneStepIntegration = compute_dt(nStep % 2) + godunov_unsplit(nStep, dt)
// (HydroRunGodunov.cpp:4082-4126).  Overrides virtual methods only (HydroRunBase.h:80,405,425,433,512).
#ifndef HYDRO_RUN_GODUNOV_HIP_H_
#define HYDRO_RUN_GODUNOV_HIP_H_

#include "HydroRunGodunov.h"
#include "RgpuBinding.h"

namespace hydroSimu {

class HydroRunGodunovHip : public HydroRunGodunov {
 public:
  explicit HydroRunGodunovHip(ConfigMap& cfg) : HydroRunGodunov(cfg), ctx_(0) {
    rgpu_params p;
    // [hydro] unsplitVersion as HydroRunGodunov reads it (HydroRunGodunov.cpp:80-84); traceVersion-independent here
    RgpuBinding::fill_params(*this, 0, configMap.getInteger("hydro", "unsplitVersion", 1), false, &p);
    RgpuBinding::check(ctx_, rgpu_create(&p, &ctx_));
  }
  virtual ~HydroRunGodunovHip() { rgpu_destroy(ctx_); }

  virtual int init_simulation(const std::string problemName) {
    const int step = HydroRunGodunov::init_simulation(problemName);
    RgpuBinding::check(ctx_, rgpu_upload(ctx_, h_U.data(), /*both=*/1));
    if (gravityEnabled) RgpuBinding::check(ctx_, rgpu_set_gravity_field(ctx_, h_gravity.data()));
    if (randomForcingEnabled) RgpuBinding::check(ctx_, rgpu_set_forcing_field(ctx_, h_randomForcing.data()));
    return step;
  }
  virtual void make_all_boundaries(HostArray<real_t>&) {
    RgpuBinding::check(ctx_, rgpu_make_all_boundaries(ctx_, 0, totalTime, 0.0));
  }
  virtual real_t compute_dt(int useU = 0) { return rgpu_compute_dt(ctx_, useU); }      // HydroRunBase.cpp:372-426
  virtual void oneStepIntegration(int& nStep, real_t& t, real_t& dt) {
    RgpuBinding::check(ctx_, rgpu_one_step_integration(ctx_, &nStep, &t, &dt));
  }
  virtual void copyGpuToCpu(int nStep = 0) {
    RgpuBinding::check(ctx_, rgpu_download(ctx_, (nStep % 2 == 0 ? h_U : h_U2).data(), nStep % 2));
  }

 private:
  rgpu_ctx* ctx_;
};

}  // namespace hydroSimu
#endif
