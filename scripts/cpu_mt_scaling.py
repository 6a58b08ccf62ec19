#!/usr/bin/env python3
"""thread scaling of the oracle's z-slab threaded 3D MHD step (orc_run_mt) on this host: one 256^3 (or --size) MRI box"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_api import Oracle
from ramsesgpu_amd.solver import load_library
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = load_library(); O = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
ini = os.path.join(ROOT, "configs", "mhd_mri_3d.ini"); ov = "mesh.nx=%d;mesh.ny=%d;mesh.nz=%d" % (n, n, n)
p = L.params_from_ini(ini, ov); U0 = L.init_condition(ini, ov, p)
cores = len(os.sched_getaffinity(0))
print("host: %d hardware threads in the affinity mask" % cores, flush=True)
for nt in [t for t in (16, 32, 64, 128, 256) if t <= cores] + ([cores] if cores not in (16, 32, 64, 128, 256) else []):
    t0 = time.time(); O.run_mt(p, U0, 1, nt); t1 = time.time() - t0
    t0 = time.time(); O.run_mt(p, U0, 3, nt); t3 = time.time() - t0
    per = (t3 - t1) / 2
    print("%4d threads: %.2f s per step = %.2f Mcell-updates/s (set-up + first step %.1f s)" % (nt, per, n ** 3 / per / 1e6, t1), flush=True)
