"""Kernel bodies + step driver of the product sources, executed through the TEST-ONLY host emulation of the launch
layer (tests/emu/rg_backend.h, built into tests/_build/librgpu_emu.so), against the oracle and the reference's
golden fixtures.  This is the CPU-side proof that index ranges, the compact traced state, the gather order of the
update and the shearing-box remaps are right; tests/test_gpu_parity.py repeats the same checks on the HIP build."""
import numpy as np
import pytest

import parity_checks as pc
from conftest import golden_cases, ini
from ramsesgpu_amd.solver import Solver, interior

GOLDEN = sorted(golden_cases())


@pytest.mark.parametrize("name", GOLDEN)
def test_golden(name, emu_lib):
    pc.check_golden_case(emu_lib, name)


@pytest.mark.parametrize("base,ov,nsteps", pc.ORACLE_RUNS, ids=["%s[%s]" % (b, o) for b, o, _ in pc.ORACLE_RUNS])
def test_run_vs_oracle(base, ov, nsteps, emu_lib, oracle):
    pc.check_run_vs_oracle(emu_lib, oracle, base, ov, nsteps)


@pytest.mark.parametrize("base,ov,mach", pc.RANDOM_STEPS, ids=["%s[%s]" % (b, o) for b, o, _ in pc.RANDOM_STEPS])
def test_single_step_on_random_state(base, ov, mach, emu_lib, oracle):
    pc.check_single_step_random(emu_lib, oracle, base, ov, mach=mach)


@pytest.mark.parametrize("base,ov", pc.BOUNDARY_CASES, ids=["%s[%s]" % c for c in pc.BOUNDARY_CASES])
def test_boundaries_and_dt(base, ov, emu_lib, oracle):
    pc.check_boundaries(emu_lib, oracle, base, ov)
    pc.check_compute_dt(emu_lib, oracle, base, ov)


@pytest.mark.parametrize("base,ov", [("mhd_mri_3d", "mesh.nx=8;mesh.ny=12;mesh.nz=26"),
                                     ("implode3d", "mesh.nx=10;mesh.ny=10;mesh.nz=20")], ids=["mri", "implode3d"])
def test_step_core_in_plane_pieces(base, ov, emu_lib):
    pc.check_core_plane_pieces(emu_lib, base, ov)


@pytest.mark.parametrize("base,ov,nsteps", pc.HISTORY_CASES, ids=["%s[%s]" % (b, o) for b, o, _ in pc.HISTORY_CASES])
def test_history_diagnostics(base, ov, nsteps, emu_lib, oracle):
    pc.check_history(emu_lib, oracle, base, ov, nsteps)


def test_run_driver_inertial_wave_history_file(emu_lib, tmp_path):
    """host logic of the run driver (time loop, history cadence, the probe row of history_inertial_wave) on the 2D
    rotating-frame step, through the emulation build: the file equals the reference's character for character"""
    import ctypes as C
    import os
    from conftest import ROOT, load_golden
    case = golden_cases()["inertialwave2d_16_history"]
    want = [str(x) for x in load_golden("inertialwave2d_16_history")["history_text"]]
    ov = case["overrides"] + ";output.outputVtk=no;output.outputHdf5=no;output.outputDir=%s" % tmp_path
    err = C.create_string_buffer(512)
    mc = C.c_double(0)
    ini = os.path.join(ROOT, "configs", case["base"] + ".ini")
    n = emu_lib.lib.rgpuh_run(ini.encode(), ov.encode(), C.byref(mc), err, 512)
    assert n == 8, err.value
    got = [ln.rstrip("\n") for ln in open(tmp_path / "mhd_inertialWave_2d_history.txt") if ln.strip() and not ln.startswith("#")]
    assert got == want


@pytest.mark.parametrize("base,ov,nsteps", pc.TURB_HISTORY_CASES, ids=["%s[%s]" % (b, o) for b, o, _ in pc.TURB_HISTORY_CASES])
def test_turbulence_history(base, ov, nsteps, emu_lib, oracle):
    pc.check_history_turbulence(emu_lib, oracle, base, ov, nsteps)


@pytest.mark.parametrize("base,ov", [("mhd_BrioWu", "mesh.nx=24;mesh.ny=16"),                       # Neumann faces, 2D MHD
                                     ("mhd_BrioWu", "mesh.nx=12;mesh.ny=10;mesh.nz=8;BrioWu.direction=0;MHD.implementationVersion=4")])
def test_public_ghost_fill_invalidates_fused_dt(base, ov, emu_lib, oracle):
    pc.check_public_ghost_fill_invalidates_fused_dt(emu_lib, oracle, base, ov)


@pytest.mark.parametrize("base,ov,nsteps", [("orszag-tang", "mesh.nx=24;mesh.ny=20", 8), ("kelvin_helmholtz_gpu_2d", "mesh.nx=24;mesh.ny=16", 8),
                                            ("implode3d", "mesh.nx=8;mesh.ny=8;mesh.nz=8", 4),
                                            ("orszag-tang3d", "mesh.nx=8;mesh.ny=8;mesh.nz=8", 6),
                                            ("mhd_mri_3d", "mesh.nx=8;mesh.ny=12;mesh.nz=8;MHD.omega0=0.02", 6),
                                            ("mhd_BrioWu", "mesh.nx=12;mesh.ny=10;mesh.nz=8;BrioWu.direction=0;MHD.implementationVersion=4", 6)])
def test_run_steps_plain_loop(base, ov, nsteps, emu_lib, oracle):
    """rgpu_run_steps on the emulation backend: 2D -- no fused kernels, the plain loop of the reference; 3D -- the batch logic of the
    device-side time step with the record formed by the same expressions (csrc/step_clock_rec.h) and resolved by value.
    K steps in one call, in pieces, and with an end time inside the batch == the oracle's run"""
    p = emu_lib.params_from_ini(ini(base), ov)
    U0 = emu_lib.init_condition(ini(base), ov, p)
    ref, dts_ref, _ = oracle.run(p, U0, nsteps)
    t_ref = 0.0
    for d in dts_ref:
        t_ref += float(d)
    sv = Solver(p, emu_lib)
    try:
        sv.start(U0, 0)
        assert emu_lib.lib.rgpu_device_time_step_ready(sv.ctx, 0) == 0
        assert sv.run_steps(3) == 3 and sv.run_steps(nsteps - 3) == nsteps - 3
        assert sv.nStep == nsteps and sv.totalTime == t_ref and sv.dt == float(dts_ref[-1])
        assert np.array_equal(interior(sv.getDataHost(), p), interior(ref, p))
        assert sv.run_steps(4, tEnd=t_ref) == 0
    finally:
        sv.close()
    sv = Solver(p, emu_lib)
    try:
        sv.start(U0, 0)
        tEnd = t_ref - 0.5 * float(dts_ref[-1])
        assert sv.run_steps(nsteps + 5, tEnd) == nsteps and sv.totalTime == t_ref
        assert np.array_equal(interior(sv.getDataHost(), p), interior(ref, p))      # the no-op steps behind the end time left the state alone
        more, dts_more, _ = oracle.run(p, U0, nsteps + 1)
        sv.oneStepIntegration()                                                      # ... and its CFL maxima and ghost cells
        assert sv.dt == float(dts_more[-1]) and np.array_equal(interior(sv.getDataHost(), p), interior(more, p))
        if p.three_d:
            assert emu_lib.lib.rgpu_clock_capable(sv.ctx) == 1
    finally:
        sv.close()


NEAR_UNIFORM = [("orszag-tang", "mesh.nx=40;mesh.ny=32"), ("orszag-tang3d", "mesh.nx=20;mesh.ny=16;mesh.nz=12"), ("mhd_mri_3d", "mesh.nx=16;mesh.ny=24;mesh.nz=12")]


@pytest.mark.parametrize("eps", [0.0, 1e-15, 3e-14, 1e-12, 3e-11, 1e-8])
@pytest.mark.parametrize("base,ov", NEAR_UNIFORM, ids=[c[0] for c in NEAR_UNIFORM])
def test_alfven_selection_near_ties(base, ov, eps, emu_lib, oracle):
    """the Alfven-speed selection of the 2D HLLD edge solver at, inside and outside its margins (2^-40 on the star ratio, 2^-45 on
    the cross products): uniform magnetised flow + perturbations of relative size eps, one step, every double equal to the oracle's"""
    pc.check_single_step_near_uniform(emu_lib, oracle, base, ov, eps)


def test_alfven_selection_against_the_reference_sequence_emu(emu_lib):
    """rgpu_selftest_alfven through the emulation build (one lane per wave: every sample takes its own route): selection == reference
    sequence on 2e5 random and adversarial edge states, and both routes occur"""
    sel, ref, kind, route = pc.check_alfven_selftest(emu_lib, 200000, 11, 1e-7)
    assert sel > 0 and ref > 0
    assert (route[kind == 2] == 1).all()
    sel, ref, _, _ = pc.check_alfven_selftest(emu_lib, 50000, 12, 1e-101)
    assert sel == 0


FUSED_FILL = [("implode3d", "mesh.nx=10;mesh.ny=8;mesh.nz=12"),                                                        # reflecting walls: corners = images of images, both signs
              ("implode3d", "mesh.nx=9;mesh.ny=7;mesh.nz=12;mesh.boundary_xmin=2;mesh.boundary_xmax=3;mesh.boundary_ymin=3;mesh.boundary_ymax=1"),   # mixed (x: not a pair of equals)
              ("orszag-tang3d", "mesh.nx=8;mesh.ny=10;mesh.nz=12"),                                                     # periodic MHD
              ("orszag-tang3d", "mesh.nx=8;mesh.ny=10;mesh.nz=12;mesh.boundary_xmin=2;mesh.boundary_xmax=2;mesh.boundary_ymin=1;mesh.boundary_ymax=2"),   # Dirichlet leaves B alone
              ("mhd_mri_3d", "mesh.nx=8;mesh.ny=12;mesh.nz=12;MHD.omega0=0.05"),                                        # shearing box: remap at (i, wrapped row), kept Bx face
              ("mhd_mri_3d", "mesh.nx=6;mesh.ny=6;mesh.nz=12;MHD.omega0=0.3")]                                          # ... shift of several cells, ny = 2 gw


@pytest.mark.parametrize("base,ov", FUSED_FILL, ids=["%s-%d" % (c[0], n) for n, c in enumerate(FUSED_FILL)])
def test_one_launch_ghost_fill_equals_the_separate_passes(base, ov, emu_lib, oracle):
    pc.check_fused_fill(emu_lib, oracle, base, ov)
