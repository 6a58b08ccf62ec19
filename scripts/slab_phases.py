#!/usr/bin/env python3
"""Phase timers of one rank of the z-slab schedule at N = 8 (a 512 x 512 x 64 slab, its own z neighbour) through the C++
driver: where a rank's step goes (sweep / update / ghost fill; no stand-alone CFL scan: it rides in the update kernels)."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
from ramsesgpu_amd import comm as rcomm
from ramsesgpu_amd.solver import load_library
L = load_library(); CL = rcomm.load_comm_library()
ini = os.path.join(ROOT, "configs", "mhd_mri_3d.ini")
run = rcomm.CommRun(ini, "mesh.nx=512;mesh.ny=512;mesh.nz=64", 0, 1, rcomm.unique_id(CL), library=L, comm_library=CL, overlap=True)
run.init_simulation()
for _ in range(3): run.oneStepIntegration()
run.solver.synchronize()
sv = run.solver
sv.enable_timers(True); sv.reset_timers()
n = 5
t0 = time.time()
for _ in range(n): run.oneStepIntegration()
sv.synchronize(); w = (time.time() - t0) / n * 1e3
tm = sv.timers()
print("with timers %.2f ms/step; phases ms/step: " % w + "  ".join("%s=%.3f" % (k, v / n * 1e3) for k, v in tm.items() if v > 0) + "  sum=%.3f" % (sum(tm.values()) / n * 1e3))
run.close()
