"""ramsesgpu_amd -- MI355X-native Godunov / MUSCL-Hancock unsplit step (hydro + MHD) behind ramsesGPU's
oneStepIntegration interface.  The product is the C-ABI shared library built from csrc/ (HIP, gfx950); this
package is the thin Python plumbing used by tests, bench.py and the multi-GPU slab driver."""
from ._capi import RgpuParams  # noqa: F401
from .solver import Library, Solver, load_library, lib_path  # noqa: F401

__all__ = ["RgpuParams", "Library", "Solver", "load_library", "lib_path"]
