#!/usr/bin/env python3
"""Per-rank cost of the z-slab schedule on ONE GPU: one rank of a ring of one (periodic z: the rank is its own neighbour,
the halo planes really travel through RCCL send / recv on the halo stream, device-local instead of over xGMI) stepping a
512 x 512 x (512/N) box through the C++ driver (include/rgpu_comm.h) with the overlapped and the serial schedule -- what a
rank of bench.py --gpus N does per step, minus the link time of 2 x 51.5 MB (~0.4 ms at 150 GB/s, hidden behind the
inner planes by the overlapped schedule)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ramsesgpu_amd import comm as rcomm
from ramsesgpu_amd.solver import load_library
L = load_library()
CL = rcomm.load_comm_library(rcomm.comm_lib_path(L.arithmetic))   # the driver built for this arithmetic (RGPU_ARITH)
ini = os.path.join(ROOT, "configs", "mhd_mri_3d.ini")
cid = rcomm.unique_id(CL)
for nz in ([int(os.environ['PROBE_NZ'])] if os.environ.get('PROBE_NZ') else (512, 256, 128, 64)):   # PROBE_NZ: one slab thickness (for rocprofv3)
    for overlap in ((True,) if os.environ.get('PROBE_NZ') else (True, False)):
        run = rcomm.CommRun(ini, "mesh.nx=512;mesh.ny=512;mesh.nz=%d" % nz, 0, 1, cid, library=L, comm_library=CL, overlap=overlap)
        run.init_simulation()
        for _ in range(3): run.oneStepIntegration()
        run.solver.synchronize(); t0 = time.time()
        n = 10
        for _ in range(n): run.oneStepIntegration()
        run.solver.synchronize(); dt = (time.time() - t0) / n
        print("nz=%3d (N=%d) %-8s %7.2f ms/step  -> %6.0f Mcell/s per rank, x%d = %6.0f" % (nz, 512 // nz, "overlap" if overlap else "serial", dt * 1e3,
              512 * 512 * nz / dt / 1e6, 512 // nz, 512 * 512 * 512 / dt / 1e6), flush=True)
        run.close()
        cid = rcomm.unique_id(CL)
