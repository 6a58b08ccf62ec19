// host_params.h -- derive the rgpu_params struct from a parameter file.
// Mirrors the HydroParameters constructor (src/hydro/HydroParameters.h:166-525) and the parts of the
// HydroRunGodunov / MHDRunGodunov constructors that select the step variant (HydroRunGodunov.cpp:63-90,
// MHDRunGodunov.cpp:97-181).
#pragma once
#include <string>

#include "../../../include/rgpu.h"
#include "ini_config.h"

namespace rgpu_host {

// Run-level knobs that are not part of the device-side parameter block.
struct RunSettings {
  int nStepmax;
  double tEnd;
  int nOutput;
  int nLog;
  std::string problem;
  std::string outputDir, outputPrefix;
  bool outputVtk;
  // [run] restart / restart_filename / restart_reset_totaltime (HydroRunBase.cpp:7033-7066, MHDRunGodunov.cpp:3805, 3866-3880)
  bool outputRestart, ghostIncluded;   // [output] outputHdf5 (the reference's HDF5 file through a dlopen'ed libhdf5, else the raw restart dump), ghostIncluded
  bool restartEnabled, restartResetTotalTime;
  bool outputVtkAscii;          // [output] outputVtkAscii: the .vti as text (12 significant digits) instead of appended raw doubles
  bool outputXsm, outputNrrd;   // [output] outputXsm / outputNrrd (HydroParameters.h:478,485)
  int hdf5CompressionLevel;   // [output] outputHdf5CompressionLevel (0..9, HydroRunBase.cpp:3408-3414)
  bool restartUpscale;   // [run] restart_upscale: the restart file holds the box at half the resolution (HydroRunBase.cpp:7044-7062)
  std::string restartFilename;
};

// Fills *p for slab `slab_rank` of `slab_count` (0,1 = whole domain).  Throws std::runtime_error on
// configurations outside the implemented scope.
void params_from_config(const IniConfig& cfg, int slab_rank, int slab_count, rgpu_params* p, RunSettings* rs);

}  // namespace rgpu_host
