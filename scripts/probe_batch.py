#!/usr/bin/env python3
"""Where does a batch of device-clock steps spend its time?  One device, no communicator: a 512 x 512 x NZ MRI box stepped
(a) one oneStepIntegration per step (host turn per step), (b) rgpu_run_steps (one batch), (c) the batch by hand through the public
pieces -- rgpu_clock_open / _tick + rgpu_step_pre / core / post_a / post_b + rgpu_clock_close -- with the HOST time to queue the
steps measured apart from the time until the device has run them.  usage: probe_batch.py [nz] [steps]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ramsesgpu_amd.solver import Library, Solver, lib_path
nz = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
base = os.environ.get("PROBE_BASE", "mhd_mri_3d")
L = Library(os.environ.get("RGPU_LIB") or lib_path())
ini = os.path.join(ROOT, "configs", base + ".ini")
ov = "mesh.nx=512;mesh.ny=512;mesh.nz=%d" % nz + (";hydro.riemannSolver=hllc" if base == "implode3d" else "")
p = L.params_from_ini(ini, ov)
sv = Solver(p, L)
sv.upload(L.init_condition(ini, ov, p), both=False); sv.make_all_boundaries(0, 0.0, 0.0)
for _ in range(3): sv.oneStepIntegration()
sv.synchronize(); t0 = time.time()
for _ in range(n): sv.oneStepIntegration()
sv.synchronize(); a = (time.time() - t0) / n
sv.synchronize(); t0 = time.time()
assert sv.run_steps(n) == n
sv.synchronize(); b = (time.time() - t0) / n
lib = L.lib
for f in ("rgpu_step_pre", "rgpu_step_core", "rgpu_step_post_a", "rgpu_step_post_b"):
    getattr(lib, f).argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
lib.rgpu_clock_open.argtypes = [C.c_void_p, C.c_double, C.c_double]
lib.rgpu_clock_tick.argtypes = [C.c_void_p]
lib.rgpu_clock_close.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
assert lib.rgpu_device_time_step_ready(sv.ctx, sv.nStep % 2) == 1
sv.synchronize(); t0 = time.time()
assert lib.rgpu_clock_open(sv.ctx, sv.totalTime, float("inf")) == 0
n0 = sv.nStep
for q in range(n):
    assert lib.rgpu_clock_tick(sv.ctx) == 0
    for f in ("rgpu_step_pre", "rgpu_step_core", "rgpu_step_post_a", "rgpu_step_post_b"):
        assert getattr(lib, f)(sv.ctx, n0 + q, 0.0, 0.0) == 0
t1 = time.time()
ran, stop, t, dtl = C.c_int(0), C.c_int(0), C.c_double(sv.totalTime), C.c_double(0)
assert lib.rgpu_clock_close(sv.ctx, n0, C.byref(ran), C.byref(t), C.byref(dtl), None, C.byref(stop)) == 0 and ran.value == n
t2 = time.time()
print("%s 512x512x%d, %s: host loop %.3f ms/step | rgpu_run_steps %.3f ms/step | batch by hand: host queued %d steps in %.3f ms/step, device done after %.3f ms/step"
      % (base, nz, L.arithmetic, a * 1e3, b * 1e3, n, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3), flush=True)
sv.close()
