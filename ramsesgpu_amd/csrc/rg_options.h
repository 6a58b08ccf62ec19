// rg_options.h -- diagnostic options of librgpu.so (process-wide), set through rgpu_set_option (include/rgpu.h).
// A user needs none of them: each switches an optional fast path off, or pins a parameter of a launch plan, so that the tests can
// run a configuration both ways (tests/test_gpu_parity.py).  They replace the RGPU_NO_* / RGPU_XCD_SUB / RGPU_ZSEG / RGPU_CHUNKS
// environment switches of rounds 2-5; what is left in the environment is listed in include/rgpu.h ("Environment").
#pragma once
#include <cstring>

namespace rgpu {

struct Options {
  int spec = 1;           // kernels specialised for the solver configuration (0: the generic instantiations only)
  int ghost_images = 1;   // the fused 2D steps write the ghost images of their output (the next step's ghost fill is skipped)
  int step_clock = 1;     // the time step stays on the device between the steps of a batch (rgpu_run_steps)
  int xcd_sub = -1;       // sub-band size (cells) of the XCD-aware workgroup order of the flat kernels; -1: rgpu_create's default
  int zseg = 0;           // planes per z segment of the tiled sweeps; 0: planned per launch (tile_grid_plan)
  int chunks = -1;        // chunks of the two-stream schedule of the flat 3D MHD kernels; -1: ksize / 8; 1: one stream
};
inline Options& options() { static Options o; return o; }
inline int* option_slot(const char* name) {
  Options& o = options();
  if (!name) return 0;
  if (!std::strcmp(name, "spec")) return &o.spec;
  if (!std::strcmp(name, "ghost_images")) return &o.ghost_images;
  if (!std::strcmp(name, "step_clock")) return &o.step_clock;
  if (!std::strcmp(name, "xcd_sub")) return &o.xcd_sub;
  if (!std::strcmp(name, "zseg")) return &o.zseg;
  if (!std::strcmp(name, "chunks")) return &o.chunks;
  return 0;
}

}  // namespace rgpu
