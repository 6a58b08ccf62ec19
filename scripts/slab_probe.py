#!/usr/bin/env python3
"""Per-rank cost of the slab schedule without communication: one rank (world 1, periodic z filled locally) stepping a
512 x 512 x (512/N) box with the overlapped and the serial schedule -- an upper bound for the strong-scaling efficiency
of bench.py --gpus N."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ramsesgpu_amd.slab import SlabRun
from ramsesgpu_amd.solver import load_library
L = load_library()
ini = os.path.join(ROOT, "configs", "mhd_mri_3d.ini")
for nz in (512, 256, 128, 64):
    for overlap in (True, False):
        run = SlabRun(ini, "mesh.nx=512;mesh.ny=512;mesh.nz=%d" % nz, library=L, device="cuda:0", overlap=overlap)
        run.init_simulation()
        for _ in range(3): run.oneStepIntegration()
        torch.cuda.synchronize(); t0 = time.time()
        n = 10
        for _ in range(n): run.oneStepIntegration()
        torch.cuda.synchronize(); dt = (time.time() - t0) / n
        print("nz=%3d (N=%d) %-8s %7.2f ms/step  -> %6.0f Mcell/s per rank, x%d = %6.0f" % (nz, 512 // nz, "overlap" if overlap else "serial", dt * 1e3,
              512 * 512 * nz / dt / 1e6, 512 // nz, 512 * 512 * 512 / dt / 1e6))
        run.close()
