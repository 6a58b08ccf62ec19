import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from ramsesgpu_amd.solver import Library, Solver, lib_path
L = Library(os.environ.get('RGPU_LIB') or lib_path())
for base, ov, nst in (("jet2d_cpu", "", 200), ("kelvin_helmholtz_gpu_2d", "mesh.nx=512;mesh.ny=512", 200), ("kelvin_helmholtz_gpu_2d", "mesh.nx=4096;mesh.ny=4096", 20), ("orszag-tang", "mesh.nx=512;mesh.ny=512", 500), ("orszag-tang", "mesh.nx=4096;mesh.ny=4096", 20)):
    ini = os.path.join(ROOT, "configs", base + ".ini")
    p = L.params_from_ini(ini, ov)
    U0 = L.init_condition(ini, ov, p)
    sv = Solver(p, L)
    sv.upload(U0, both=False); sv.make_all_boundaries(0, 0.0, 0.0); sv.upload(sv.getDataHost(0), both=True)
    for _ in range(5): sv.oneStepIntegration()
    sv.synchronize(); t0 = time.time()
    for _ in range(nst): sv.oneStepIntegration()
    sv.synchronize(); dt = (time.time() - t0) / nst
    sv.synchronize(); t0 = time.time()
    assert sv.run_steps(nst) == nst       # the same steps as one call of the product's loop (device-side time step where it applies)
    sv.synchronize(); dtb = (time.time() - t0) / nst
    clocked = L.lib.rgpu_device_time_step_ready(sv.ctx, sv.nStep % 2)
    sv.enable_timers(True); sv.reset_timers()
    for _ in range(5): sv.oneStepIntegration()
    tm = sv.timers()
    print("%-26s %-28s %9.1f Mcell/s %8.4f ms/step | run_steps%s %9.1f Mcell/s %8.4f ms/step  " % (base, ov, p.nx * p.ny / dt / 1e6, dt * 1e3, "(device dt)" if clocked else "(plain loop)", p.nx * p.ny / dtb / 1e6, dtb * 1e3) + " ".join("%s=%.4f" % (k, v / 5 * 1e3) for k, v in tm.items() if v > 0), flush=True)
    sv.close()
