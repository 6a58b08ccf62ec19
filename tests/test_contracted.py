"""The opt-in "contracted arithmetic" variant of the product (ramsesgpu_amd/librgpu_fast.so: the same sources built with FMA
contraction and ~1-ulp division / square root, see rgpu_arithmetic() in include/rgpu.h) against the reference's golden
fixtures and against the exact library: agreement to round-off -- the stated tolerance, relative L2 < 1e-12 per run
(parity_checks.L2_TOLERANCE; measured worst over the fixtures: 2e-14) -- instead of equal bits."""
import os
import subprocess
import sys

import numpy as np
import pytest

import parity_checks as pc
from conftest import ROOT, golden_cases, ini

pytestmark = pytest.mark.gpu
GOLDEN = sorted(golden_cases())


@pytest.mark.parametrize("name", GOLDEN)
def test_golden_within_tolerance(name, gpu_contracted_lib):
    pc.check_golden_case(gpu_contracted_lib, name, exact=False)


@pytest.mark.parametrize("base,ov,nsteps", [
    ("mhd_mri_3d", "mesh.nx=32;mesh.ny=64;mesh.nz=32", 20),                       # tiled MHD sweep, several tiles and segments
    ("implode3d", "mesh.nx=48;mesh.ny=48;mesh.nz=48;hydro.riemannSolver=hllc", 20),  # tiled hydro sweep
    ("orszag-tang", "mesh.nx=128;mesh.ny=128", 50),
])
def test_contracted_against_exact(base, ov, nsteps, gpu_lib, gpu_contracted_lib):
    from ramsesgpu_amd.solver import Solver, interior
    out, dts = [], []
    for lib in (gpu_lib, gpu_contracted_lib):
        p = lib.params_from_ini(ini(base), ov)
        U0 = lib.init_condition(ini(base), ov, p)
        sv = Solver(p, lib)
        try:
            dts.append(np.array(sv.start(U0, nsteps)))
            out.append(interior(sv.getDataHost(), p).copy())
        finally:
            sv.close()
    err = pc.rel_l2(out[1], out[0])
    assert np.isfinite(out[1]).all() and err < pc.L2_TOLERANCE, "%s: relative L2 %.3e" % (base, err)
    assert np.abs(dts[1] / dts[0] - 1.0).max() < 1e-11
    assert not np.array_equal(out[0], out[1]), "the contracted variant returned the exact library's bits: is it the right build?"


def test_contracted_slab_driver_equals_the_single_device_run():
    """librgpu_comm_fast.so (one rank, its own z neighbour, halo planes through RCCL) against librgpu_fast.so alone: same
    arithmetic, hence the same bits.  In a subprocess: the two variants export the same symbols, a process holds one."""
    code = r'''
import os, sys
import numpy as np
sys.path.insert(0, %r)
from ramsesgpu_amd import comm as rcomm
from ramsesgpu_amd.solver import Library, Solver, interior, lib_path
L = Library(lib_path("contracted")); assert L.arithmetic == "contracted"
CL = rcomm.load_comm_library(rcomm.comm_lib_path("contracted"))
ini = os.path.join(%r, "configs", "mhd_mri_3d.ini"); ov = "mesh.nx=16;mesh.ny=32;mesh.nz=16"
run = rcomm.CommRun(ini, ov, 0, 1, rcomm.unique_id(CL), library=L, comm_library=CL)
run.init_simulation()
for _ in range(4): run.oneStepIntegration()
a = run.local_interior().copy(); run.close()
p = L.params_from_ini(ini, ov); sv = Solver(p, L); sv.start(L.init_condition(ini, ov, p), 4)
b = interior(sv.getDataHost(), p); sv.close()
assert np.array_equal(a, b), int((a != b).sum())
print("OK")
''' % (ROOT, ROOT)
    env = dict(os.environ); env.pop("RGPU_LIB", None); env.pop("RGPU_ARITH", None)
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-2000:]
