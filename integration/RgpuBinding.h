// RgpuBinding.h -- reference-side helper shared by MHDRunGodunovHip.h and HydroRunGodunovHip.h: fills rgpu_params from
// the members of a ramsesGPU run object (HydroParameters.h:73-136 and the derived values HydroParameters.h:196-325 has
// already stored in _gParams) and turns RGPU_E* codes into exceptions.  Compiles with g++ -std=c++11 -DUSE_DOUBLE
// against the UNMODIFIED reference headers (tests/test_integration_stub.py); needs no HIP header -- the ABI is plain C.
#ifndef RGPU_BINDING_H_
#define RGPU_BINDING_H_

#include <stdexcept>
#include <string>

#include "HydroRunBase.h"
extern "C" {
#include "rgpu.h"
}

namespace hydroSimu {

// RUN is a HydroRunBase subclass; the protected members are reached through a derived accessor
struct RgpuBinding : public HydroRunBase {
  static void fill_params(HydroRunBase& base, int implementationVersion, int unsplitVersion, bool shearingBox, rgpu_params* out) {
    RgpuBinding& r = static_cast<RgpuBinding&>(base);   // layout-compatible view: no data members added
    rgpu_params p = rgpu_params();
    p.abi_version = RGPU_ABI_VERSION;
    p.nx = r.nx; p.ny = r.ny; p.nz = r.nz; p.nz_global = r.nz; p.slab_rank = 0; p.slab_count = 1;
    p.ghostWidth = r.ghostWidth; p.nbVar = r.nbVar; p.mhdEnabled = r.mhdEnabled ? 1 : 0;
    p.bc[0] = r.boundary_xmin; p.bc[1] = r.boundary_xmax; p.bc[2] = r.boundary_ymin;
    p.bc[3] = r.boundary_ymax; p.bc[4] = r.boundary_zmin; p.bc[5] = r.boundary_zmax;
    p.xMin = r._gParams.xMin; p.xMax = r._gParams.xMax; p.yMin = r._gParams.yMin; p.yMax = r._gParams.yMax;
    p.zMin = r._gParams.zMin; p.zMax = r._gParams.zMax; p.dx = r._gParams.dx; p.dy = r._gParams.dy; p.dz = r._gParams.dz;
    p.cfl = r.cfl; p.gamma0 = r._gParams.gamma0; p.cIso = r._gParams.cIso; p.smallr = r._gParams.smallr;
    p.smallc = r._gParams.smallc; p.smalle = r._gParams.smalle; p.smallp = r._gParams.smallp;
    p.smallpp = r._gParams.smallpp; p.gamma6 = r._gParams.gamma6; p.Omega0 = r._gParams.Omega0;
    p.slope_type = r._gParams.slope_type; p.niter_riemann = r._gParams.niter_riemann; p.iorder = r._gParams.iorder;
    p.riemannSolver = r._gParams.riemannSolver; p.magRiemannSolver = r._gParams.magRiemannSolver;
    p.implementationVersion = implementationVersion; p.unsplitVersion = unsplitVersion;
    p.shearingBoxEnabled = shearingBox ? 1 : 0;
    p.enableJet = r.enableJet; p.ijet = r.ijet; p.offsetJet = r.offsetJet;
    p.djet = r.djet; p.ujet = r.ujet; p.pjet = r.pjet; p.cjet = r.cjet;
    p.gravityEnabled = r.gravityEnabled ? 2 : 0;   // 2: per-cell h_gravity, uploaded in init_simulation
                                                   // (1 + gravity_x/y/z: a uniform vector without the array)
    p.nu = r._gParams.nu; p.eta = r._gParams.eta;  // dissipative stage inside rgpu_godunov_unsplit
    p.zStratifiedFloor = r.configMap.getBool("MRI", "floor", false) ? 1 : 0;   // BC_Z_STRATIFIED (HydroRunBase.cpp:2206)
    p.randomForcingEnabled = r.randomForcingEnabled ? 1 : 0; p.randomForcingEdot = r.randomForcingEdot;
    // problem "turbulence-Ornstein-Uhlenbeck": the library owns the process (same parameters as ForcingOrnsteinUhlenbeck reads)
    p.ouForcingEnabled = r.randomForcingOrnsteinUhlenbeckEnabled ? 1 : 0;
    p.ouInitRandom = r.configMap.getInteger("turbulence-Ornstein-Uhlenbeck", "init_random", 600);
    p.ouTimeScaleTurb = r.configMap.getFloat("turbulence-Ornstein-Uhlenbeck", "timeScaleTurb", 0.1);
    p.ouAmplitudeTurb = r.configMap.getFloat("turbulence-Ornstein-Uhlenbeck", "amplitudeTurb", 0.0001);
    p.ouKsi = r.configMap.getFloat("turbulence-Ornstein-Uhlenbeck", "ksi", 0.0);
    *out = p;
  }
  static void check(rgpu_ctx* ctx, int rc) {
    if (rc != RGPU_OK) throw std::runtime_error(std::string("librgpu: ") + rgpu_last_error(ctx));
  }
 private:
  RgpuBinding();   // never constructed
};

}  // namespace hydroSimu
#endif
