"""Host logic: parameter-file semantics of ConfigMap/HydroParameters and the slab-aware initial conditions."""
import ctypes as C

import numpy as np
import pytest

from conftest import ini
from ramsesgpu_amd import _capi


def test_float_parsed_knobs_and_derivations(product_lib):
    p = product_lib.params_from_ini(ini("orszag-tang"))
    # every real knob goes through strtof (ConfigMap.cpp:41-49)
    assert p.gamma0 == float(np.float32(1.666)) == 1.66600000858306884765625
    assert p.cfl == float(np.float32(0.4))
    assert p.smallr == float(np.float32(1e-7))
    assert p.smallp == p.smallc * p.smallc / p.gamma0
    assert p.smallpp == p.smallr * p.smallp
    assert p.gamma6 == (p.gamma0 + 1.0) / (2.0 * p.gamma0)
    assert (p.nx, p.ny, p.nz, p.nbVar, p.ghostWidth, p.mhdEnabled) == (512, 512, 1, 8, 3, 1)   # MHD forces gw=3, nbVar=8
    assert p.dx == 1.0 / 512 and p.riemannSolver == _capi.RS_HLLD and p.implementationVersion == 1
    assert list(p.bc) == [3] * 6 and not p.three_d


def test_isothermal_and_shearing_box(product_lib):
    p = product_lib.params_from_ini(ini("mhd_mri_3d"))
    assert p.cIso == float(np.float32(0.001)) and p.smallp == p.smallr * p.cIso * p.cIso
    assert p.shearingBoxEnabled == 1 and p.Omega0 == float(np.float32(0.001))
    assert (p.xMin, p.xMax, p.yMin, p.yMax) == (-0.5, 0.5, -2.0, 2.0) and p.dy == 4.0 / 32
    assert list(p.bc)[:2] == [4, 4]


def test_hydro_defaults_and_jet(product_lib):
    p = product_lib.params_from_ini(ini("jet2d_cpu"))
    assert (p.nbVar, p.ghostWidth, p.enableJet, p.ijet, p.offsetJet) == (4, 2, 1, 10, 10)
    assert p.cjet == np.sqrt(p.gamma0 * p.pjet / p.djet) and p.gamma0 == float(np.float32(1.4))
    assert p.riemannSolver == _capi.RS_APPROX
    q = product_lib.params_from_ini(ini("implode3d"), "hydro.riemannSolver=NoSuchSolver")
    assert q.riemannSolver == _capi.RS_APPROX       # unknown names silently fall back (HydroParameters.h:353-381)
    q = product_lib.params_from_ini(ini("implode3d"), "hydro.riemannSolver=HLLC")
    assert q.riemannSolver == _capi.RS_HLLC         # value is lower-cased first


def test_ini_syntax(product_lib, tmp_path):
    f = tmp_path / "t.ini"
    f.write_text("# comment\n; comment\n[Mesh]\nNX = 24 ; inline comment\nny=0x10\nnz=1\n[HYDRO]\nproblem=implode\ncfl=0,9\n"
                 "gamma0 = 1.4;not a comment\n[output]\noutputVtk=on\n")
    p = product_lib.params_from_ini(str(f))
    assert (p.nx, p.ny) == (24, 16)                 # keys case-insensitive, strtol base 0
    assert p.cfl == 0.5                             # "0,9" parses as 0 -> reset to 0.5 (HydroParameters.h:279-282)
    assert p.gamma0 == float(np.float32(1.4))


def test_slab_params(product_lib):
    full = product_lib.params_from_ini(ini("mhd_mri_3d"))
    for r in range(4):
        p = product_lib.params_from_ini(ini("mhd_mri_3d"), "", slab=(r, 4))
        assert (p.nz, p.nz_global, p.slab_rank, p.slab_count) == (4, 16, r, 4) and p.dz == full.dz
        assert p.bc[4] == _capi.BC_COPY and p.bc[5] == _capi.BC_COPY          # periodic z -> ring
    lo = product_lib.params_from_ini(ini("implode3d"), "", slab=(0, 2))
    hi = product_lib.params_from_ini(ini("implode3d"), "", slab=(1, 2))
    assert (lo.bc[4], lo.bc[5], hi.bc[4], hi.bc[5]) == (1, _capi.BC_COPY, _capi.BC_COPY, 1)


@pytest.mark.parametrize("base,ov", [("mhd_mri_3d", ""), ("orszag-tang3d", "mesh.nx=12;mesh.ny=10;mesh.nz=8;OrszagTang.kt=1.0"),
                                     ("implode3d", "mesh.nx=8;mesh.ny=8;mesh.nz=8")])
def test_slab_initial_condition_equals_planes_of_the_full_one(product_lib, base, ov):
    """incl. the skip-ahead of the drand48 stream (MRI draws 4 numbers per cell in k,j,i order, ghosts included)"""
    L = product_lib
    full_p = L.params_from_ini(ini(base), ov)
    full = L.init_condition(ini(base), ov, full_p)
    gw, n = full_p.ghostWidth, 2
    for r in range(n):
        p = L.params_from_ini(ini(base), ov, slab=(r, n))
        part = L.init_condition(ini(base), ov, p)
        k0 = r * p.nz
        ref = full[:, k0:k0 + p.nz + 2 * gw]
        if base == "implode3d":   # hydro ICs fill the interior only: compare interior planes
            assert np.array_equal(part[:, gw:-gw, gw:-gw, gw:-gw], ref[:, gw:-gw, gw:-gw, gw:-gw])
        elif base == "orszag-tang3d":
            # the energy needs the +1 neighbours in x,y only -> all planes comparable
            assert np.array_equal(part, ref)
        else:
            assert np.array_equal(part, ref)


def test_drand48_stream_matches_libc(product_lib):
    """MRI density_fluctuations uses the first draw of every cell: compare with glibc's drand48"""
    L = product_lib
    ov = "mesh.nx=4;mesh.ny=4;mesh.nz=4;MRI.density_fluctuations=0.5;MRI.seed=7"
    p = L.params_from_ini(ini("mhd_mri_3d"), ov)
    U = L.init_condition(ini("mhd_mri_3d"), ov, p)
    libc = C.CDLL("libc.so.6")
    libc.drand48.restype = C.c_double
    libc.srand48(7)
    draws = np.array([libc.drand48() for _ in range(4 * U[0].size)]).reshape(-1, 4)
    d0, d_amp = 1.0, 0.5
    assert np.array_equal(U[0].ravel(), d0 * (1 + d_amp * 2 * (draws[:, 0] - 0.5)))
