#!/usr/bin/env python3
"""GPU probe of the LDS-tiled sweeps: ms/step and phase timers of one workload with the tiled kernels on / off.
usage: probe_sweep.py implode3d 256 [steps]   |   probe_sweep.py mhd_mri_3d 512   |   probe_sweep.py orszag-tang3d 256
RGPU_TILED=0: the flat kernels; PROBE_OPTIONS: diagnostic options of the library; RGPU_LIB: another build of it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ramsesgpu_amd.solver import Library, Solver, lib_path

args = [a for a in sys.argv[1:] if not a.startswith('--')]
base = args[0]; n = int(args[1]); nst = int(args[2]) if len(args) > 2 else 10
ov = "mesh.nx=%d;mesh.ny=%d;mesh.nz=%d" % (n, n, int(os.environ.get("PROBE_NZ", n)))
if base == "implode3d":
    ov += ";hydro.riemannSolver=hllc"
L = Library(os.environ.get('RGPU_LIB') or lib_path())   # RGPU_LIB: an experiment build of the library
for kv in filter(None, os.environ.get("PROBE_OPTIONS", "").split(",")):   # PROBE_OPTIONS="zseg=64,spec=0": diagnostic options of the library
    L.set_option(kv.split("=")[0], int(kv.split("=")[1]))
ini = os.path.join(ROOT, "configs", base + ".ini")
p = L.params_from_ini(ini, ov)
U0 = L.init_condition(ini, ov, p)
sv = Solver(p, L)
sv.upload(U0, both=False); sv.make_all_boundaries(0, 0.0, 0.0); sv.upload(sv.getDataHost(0), both=True)
del U0
for _ in range(3): sv.oneStepIntegration()
sv.synchronize()
t0 = time.time()
for _ in range(nst): sv.oneStepIntegration()
sv.synchronize(); dtw = time.time() - t0
tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("RGPU_"))
print("%-14s %4d^3 [%s] %8.1f Mcell/s  %.3f ms/step" % (base, n, tag, nst * n ** 3 / dtw / 1e6, dtw / nst * 1e3), flush=True)
sv.enable_timers(True); sv.reset_timers()
for _ in range(3): sv.oneStepIntegration()
tm = sv.timers()
print("     phases ms/step: " + "  ".join("%s=%.3f" % (k, v / 3 * 1e3) for k, v in tm.items() if v > 0) + "   sum=%.3f" % (sum(tm.values()) / 3 * 1e3), flush=True)
sv.close()
