import os, sys, time, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
from ramsesgpu_amd.solver import Solver, load_library
L = load_library(); ini = os.path.join(ROOT, "configs", "mhd_mri_3d.ini")
ov = "mesh.nx=256;mesh.ny=256;mesh.nz=256"
p = L.params_from_ini(ini, ov); U0 = L.init_condition(ini, ov, p)
sv = Solver(p, L); sv.upload(U0, both=False); sv.make_all_boundaries(0, 0.0, 0.0); sv.upload(sv.getDataHost(0), both=True)
h0 = sv.history_mri()
t0 = time.time()
for n in range(300): sv.oneStepIntegration()
sv.synchronize(); el = time.time() - t0
h1 = sv.history_mri()
print("300 steps 256^3: %.1f ms/step, t=%.4g" % (el / 300 * 1e3, sv.totalTime))
print("mass", h0["mass"], h1["mass"], "rel drift", abs(h1["mass"] - h0["mass"]) / h0["mass"])
print("divB sum", h1["divB"], "maxwell", h1["maxwell"], "reynolds", h1["reynolds"], "magp", h1["magp"])
U = sv.getDataHost(); print("finite", np.isfinite(U).all(), "rho min/max", U[0].min(), U[0].max())
