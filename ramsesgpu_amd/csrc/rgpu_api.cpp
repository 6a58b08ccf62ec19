// rgpu_api.cpp -- implementation of the C ABI of include/rgpu.h: context, step driver, ghost fill, CFL scan.
//
// Compiled by hipcc (-x hip --offload-arch=gfx950 -ffp-contract=off).  All work of a context is issued on one
// HIP stream in program order; the only host<->device traffic inside the path is the 8-byte result of the CFL
// reduction (the reference copies <=192 partial maxima per step, MHDRunBase.cpp:103-128).
#include "../../include/rgpu.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

#include "launchers.h"
#include "rg_options.h"
#include "rg_tiled.h"   // LDS-tiled cooperative kernels of the backend (hip/rg_tiled.h)

using namespace rgpu;
using namespace rgpu_dev;

#include "api/ctx.h"
#include "api/boundaries.h"
#include "api/step.h"
#include "api/history.h"
#include "api/entry_core.h"
#include "api/entry_clock.h"
#include "api/entry_misc.h"
