#!/usr/bin/env python3
"""GPU probe: parity of librgpu.so against the committed golden fixtures + phase timings at growing sizes."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ramsesgpu_amd.solver import Library, Solver, interior, lib_path

L = Library(lib_path())
print("backend", L.backend)
cases = json.load(open(os.path.join(ROOT, "tests/golden/cases.json")))
nbad = 0
if "--no-parity" not in sys.argv:
    for name in sorted(cases):
        cs = cases[name]
        ini = os.path.join(ROOT, "configs", cs["base"] + ".ini")
        p = L.params_from_ini(ini, cs["overrides"])
        g = np.load(os.path.join(ROOT, "tests/golden", name + ".npz"))
        for s in cs["steps"]:
            U0 = L.init_condition(ini, cs["overrides"], p)
            sv = Solver(p, L)
            sv.start(U0, s)
            inner = interior(sv.getDataHost(), p); ref = g["step_%d" % s]
            bad = int((inner != ref).sum()); nbad += bad
            rel = np.sqrt(((inner - ref) ** 2).sum() / max((ref ** 2).sum(), 1e-300))
            print("%-28s step %3d  mismatching=%d/%d maxabs=%.3e relL2=%.3e" % (name, s, bad, ref.size, np.abs(inner - ref).max(), rel))
            sv.close()
    print("TOTAL mismatching doubles:", nbad)

sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [128, 256]
for n in sizes:
    for base, ov in (("mhd_mri_3d", "mesh.nx=%d;mesh.ny=%d;mesh.nz=%d" % (n, n, n)),
                     ("orszag-tang3d", "mesh.nx=%d;mesh.ny=%d;mesh.nz=%d" % (n, n, n)),
                     ("implode3d", "mesh.nx=%d;mesh.ny=%d;mesh.nz=%d;hydro.riemannSolver=hllc" % (n, n, n))):
        if n > 256 and base != "mhd_mri_3d":
            continue
        ini = os.path.join(ROOT, "configs", base + ".ini")
        p = L.params_from_ini(ini, ov)
        t0 = time.time(); U0 = L.init_condition(ini, ov, p); t_ic = time.time() - t0
        sv = Solver(p, L)
        sv.upload(U0, both=False); sv.make_all_boundaries(0, 0.0, 0.0); sv.upload(sv.getDataHost(0), both=True)
        del U0
        for _ in range(2): sv.oneStepIntegration()
        sv.synchronize(); nst = 5
        t0 = time.time()
        for _ in range(nst): sv.oneStepIntegration()
        sv.synchronize(); dtw = time.time() - t0
        print("%-14s %4d^3  %8.1f Mcell/s  (%.2f ms/step, IC %.1fs, dev %.1f GB)" % (base, n, nst * n ** 3 / dtw / 1e6, dtw / nst * 1e3, t_ic, L.lib.rgpu_device_bytes(p) / 1e9))
        sv.enable_timers(True); sv.reset_timers()
        for _ in range(3): sv.oneStepIntegration()
        tm = sv.timers(); tot = sum(tm.values())
        print("     phases ms/step: " + "  ".join("%s=%.2f" % (k, v / 3 * 1e3) for k, v in tm.items() if v > 0) + "   sum=%.2f" % (tot / 3 * 1e3))
        sv.close()
