// rg_tiled.h (HIP / gfx950) -- cooperative LDS-tiled kernels of the backend.  The step driver calls
//   rgpu_tiled::hydro3d_sweep(...)      whole 3D hydro step for a plane range
//   rgpu_tiled::mhd3d_sweep(...)        trace + Riemann problems of the 3D MHD step for a plane range
//   rgpu_tiled::mhd2d_step(...)         whole 2D MHD step
//   rgpu_tiled::hydro2d_step(...)       whole 2D hydro step
// each returning 0 = done, 1 = configuration not covered (the driver runs the flat per-cell kernels), < 0 = error.
#pragma once
#include "tiled_hydro.h"
#include "tiled_mhd.h"
#include "tiled_mhd2d.h"
#include "tiled_hydro2d.h"
