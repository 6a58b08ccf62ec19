"""Parity checks shared by the CPU emulation tests (tests/test_kernels_emu.py, not gpu) and the real thing
(tests/test_gpu_parity.py, -m gpu).  Every check drives a library through the C ABI (ramsesgpu_amd.solver) and
compares with the oracle (tests/oracle_api.py) or with fixtures produced by the reference binary.

Bar (stated once, used everywhere): BIT-IDENTICAL doubles.  The path is fp64; north_star asks for
"Orszag-Tang L2 error vs euler_cpu < 1e-12" -- `assert_same` reports that relative L2 too, and fails on the
stricter bit-identity."""
import numpy as np

from conftest import golden_cases, ini, load_golden
from ramsesgpu_amd import _capi
from ramsesgpu_amd.solver import Solver, interior

L2_TOLERANCE = 1e-12   # north_star's tolerance; the tests demand 0


def rel_l2(a, b):
    num = np.sqrt(((a - b) ** 2).sum())
    den = np.sqrt((b ** 2).sum())
    return num / den if den > 0 else num


def assert_same(got, ref, what, exact=True):
    """bit-identical, except (exact=False) for runs with (a) the random forcing of the "turbulence" problem: its
    normalisation is a sum over the whole domain that the reference accumulates sequentially and the device in a
    fixed parallel order; (b) the Ornstein-Uhlenbeck forcing: 31 cos() per cell from the device's libm instead of
    glibc's.  Those agree to round-off -- well inside the stated L2 tolerance -- not bit for bit"""
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert np.isfinite(got).all(), "%s: non-finite values" % what
    err = rel_l2(got, ref)
    assert err < L2_TOLERANCE, "%s: relative L2 %.3e exceeds the stated tolerance %.0e" % (what, err, L2_TOLERANCE)
    if not exact:
        return
    nbad = int((got != ref).sum())
    assert nbad == 0, "%s: %d of %d doubles differ (max abs %.3e, rel L2 %.3e)" % (what, nbad, ref.size, np.abs(got - ref).max(), err)


def attach_gravity(lib, base, ov, p, sv=None, oracle=None):
    """per-cell gravity field of the problem (params.gravityEnabled == 2), handed to the solver and / or the oracle"""
    G = lib.init_gravity(ini(base), ov, p)
    if sv is not None and G is not None:
        sv.set_gravity_field(G)
    if oracle is not None:
        oracle.set_gravity_field(G)
    F = lib.init_forcing(ini(base), ov, p)   # and the static driving field of the "turbulence" problem
    if sv is not None and F is not None:
        sv.set_forcing_field(F)
    if oracle is not None:
        oracle.set_forcing_field(F)
    return G


# ---- 1. fixtures produced by the reference binary --------------------------------------------------------------
def check_golden_case(lib, name, exact=True):
    """exact=False: the contracted-arithmetic variant of the library -- the stated L2 tolerance instead of equal bits"""
    case = golden_cases()[name]
    p = lib.params_from_ini(ini(case["base"]), case["overrides"])
    g = load_golden(name)
    for s in case["steps"]:
        U0 = lib.init_condition(ini(case["base"]), case["overrides"], p)
        sv = Solver(p, lib)
        try:
            attach_gravity(lib, case["base"], case["overrides"], p, sv=sv)
            sv.start(U0, s)
            assert_same(interior(sv.getDataHost(), p), g["step_%d" % s], "%s step %d vs reference" % (name, s),
                        exact=exact and not (p.randomForcingEnabled or p.ouForcingEnabled))
            if s == max(case["steps"]) and np.isfinite(g["total_time"]):
                assert abs(sv.totalTime - float(g["total_time"])) <= 1e-11 * max(1.0, abs(sv.totalTime))
        finally:
            sv.close()


# ---- 2. against the oracle on the same seeded inputs ---------------------------------------------------------
def random_state(p, seed, mach=1.5):
    """a physically admissible but rough state (exercises limiter branches, supersonic Riemann fans, floors)"""
    rng = np.random.RandomState(seed)
    nv, ks, js, is_ = p.shape
    rho = 0.5 + 1.5 * rng.rand(ks, js, is_)
    vel = mach * (2 * rng.rand(3, ks, js, is_) - 1)
    pres = 0.2 + rng.rand(ks, js, is_)
    U = np.zeros(p.shape)
    U[0] = rho
    U[2] = rho * vel[0]
    U[3] = rho * vel[1]
    ekin = 0.5 * rho * (vel[0] ** 2 + vel[1] ** 2)
    if nv >= 5:
        if p.three_d or nv == 8:
            U[4] = rho * vel[2]
            ekin = ekin + 0.5 * rho * vel[2] ** 2
    emag = 0.0
    if nv == 8:
        B = 0.7 * (2 * rng.rand(3, ks, js, is_) - 1)
        U[5:8] = B
        emag = 0.5 * (B ** 2).sum(0) * 1.5   # generous: cell-centred averages are smaller than face values
    U[1] = pres / (p.gamma0 - 1.0) + ekin + emag
    return U


def check_boundaries(lib, oracle, base, ov, seed=1):
    p = lib.params_from_ini(ini(base), ov)
    U = random_state(p, seed)
    sv = Solver(p, lib)
    try:
        for idim in (1, 2, 3) if p.three_d else (1, 2):
            ref = U.copy()
            oracle.make_boundaries(p, ref, idim)
            sv.upload(U, both=False)
            sv.make_boundaries(0, idim)
            assert_same(sv.getDataHost(0), ref, "%s make_boundaries(dir %d)" % (base, idim))
        ref = U.copy()
        oracle.make_all_boundaries(p, ref, 3.7, 0.9)
        sv.upload(U, both=False)
        sv.make_all_boundaries(0, 3.7, 0.9)
        assert_same(sv.getDataHost(0), ref, "%s make_all_boundaries" % base)
    finally:
        sv.close()


def check_compute_dt(lib, oracle, base, ov, seed=2):
    p = lib.params_from_ini(ini(base), ov)
    U = random_state(p, seed)
    sv = Solver(p, lib)
    try:
        sv.upload(U, both=True)
        for useU in (0, 1):
            got = sv.compute_dt(useU)
            ref = oracle.compute_dt(p, U)
            assert got == ref, "%s compute_dt(%d): %r != %r" % (base, useU, got, ref)
    finally:
        sv.close()


def check_single_step_random(lib, oracle, base, ov, seed=3, mach=1.5, t0=2.0):
    """one godunov_unsplit on a rough random state, interior (and evolved faces) compared"""
    p = lib.params_from_ini(ini(base), ov)
    U = random_state(p, seed, mach)
    oracle.make_all_boundaries(p, U, t0, 0.0)
    dt = 0.3 * oracle.compute_dt(p, U)
    sv = Solver(p, lib)
    try:
        attach_gravity(lib, base, ov, p, sv=sv, oracle=oracle)
        sv.upload(U, both=True)
        sv.godunov_unsplit(0, dt, t0)
        got = sv.getDataHost(1)
        ref = oracle.godunov_unsplit(p, U.copy(), dt, t0)
        assert_same(interior(got, p), interior(ref, p), "%s [%s] single step on random state" % (base, ov))
        if p.mhdEnabled and p.Omega0 > 0:
            # rotating path fills the ghosts of the OUTPUT at step end: the whole array is defined
            assert_same(got, ref, "%s single rotating step incl. ghosts" % base)
    finally:
        sv.close()


def near_uniform_state(p, seed, eps):
    """a uniform magnetised flow + perturbations of relative size eps in every variable: with eps around 1e-12 the candidates of the
    Alfven-speed maxima of the 2D HLLD edge solver (dev_numerics.h: alfven_pick / alfven_duel) lie within, at and beyond the margins
    below which the solver falls back to the reference's own sequence -- lanes of one wave take different routes; eps = 0: every
    candidate pair ties exactly"""
    rng = np.random.RandomState(seed)
    nv, ks, js, is_ = p.shape
    def pert():
        return 1.0 + eps * (2 * rng.rand(ks, js, is_) - 1)
    rho = 1.3 * pert()
    vel = [0.31 * pert(), -0.17 * pert(), 0.23 * pert()]
    U = np.zeros(p.shape)
    U[0] = rho
    U[2], U[3] = rho * vel[0], rho * vel[1]
    ekin = 0.5 * rho * (vel[0] ** 2 + vel[1] ** 2)
    if nv == 8 or p.three_d:
        U[4] = rho * vel[2]
        ekin = ekin + 0.5 * rho * vel[2] ** 2
    emag = 0.0
    if nv == 8:
        B = [0.45 * pert(), 0.36 * pert(), -0.52 * pert()]
        for c in range(3):
            U[5 + c] = B[c]
        emag = 0.5 * (B[0] ** 2 + B[1] ** 2 + B[2] ** 2)
    U[1] = 0.9 * pert() / (p.gamma0 - 1.0) + ekin + emag
    return U


def check_single_step_near_uniform(lib, oracle, base, ov, eps, seed=11, t0=0.0):
    """one godunov_unsplit from near_uniform_state, every double compared (the Alfven selection must return the reference's bits
    whichever route a lane takes)"""
    p = lib.params_from_ini(ini(base), ov)
    U = near_uniform_state(p, seed, eps)
    oracle.make_all_boundaries(p, U, t0, 0.0)
    dt = 0.4 * oracle.compute_dt(p, U)
    sv = Solver(p, lib)
    try:
        sv.upload(U, both=True)
        sv.godunov_unsplit(0, dt, t0)
        got = sv.getDataHost(1)
        ref = oracle.godunov_unsplit(p, U.copy(), dt, t0)
        assert_same(interior(got, p), interior(ref, p), "%s [%s] single step from a near-uniform state, eps = %g" % (base, ov, eps))
    finally:
        sv.close()


def check_run_vs_oracle(lib, oracle, base, ov, nsteps, exact=True):
    """exact=False: the contracted-arithmetic variant of the library -- relative L2 < 1e-12 on the state, dt within 1e-11"""
    p = lib.params_from_ini(ini(base), ov)
    U0 = lib.init_condition(ini(base), ov, p)
    attach_gravity(lib, base, ov, p, oracle=oracle)
    ref, dts_ref, t_ref = oracle.run(p, U0, nsteps)
    sv = Solver(p, lib)
    try:
        attach_gravity(lib, base, ov, p, sv=sv)
        dts = sv.start(U0, nsteps)
        if not exact:
            assert np.abs(np.array(dts) / np.asarray(dts_ref) - 1.0).max() < 1e-11, "%s: dt sequences differ beyond round-off" % base
            assert_same(interior(sv.getDataHost(), p), interior(ref, p), "%s [%s] %d steps vs oracle" % (base, ov, nsteps), exact=False)
            return p
        if p.randomForcingEnabled or p.ouForcingEnabled:   # round-off level agreement only (see assert_same)
            np.testing.assert_allclose(np.array(dts), dts_ref, rtol=1e-12, atol=0)
            assert_same(interior(sv.getDataHost(), p), interior(ref, p), "%s [%s] %d steps vs oracle" % (base, ov, nsteps), exact=False)
            return
        assert np.array_equal(np.array(dts), dts_ref), "%s: dt sequences differ" % base
        assert_same(interior(sv.getDataHost(), p), interior(ref, p), "%s [%s] %d steps vs oracle" % (base, ov, nsteps))
        assert sv.totalTime == t_ref
    finally:
        sv.close()
    return p


# configurations without a golden fixture: solver options / boundary types / shapes
ORACLE_RUNS = [
    # 2D MHD: other Riemann solvers, slope types, rectangular grids, smallest legal grid
    ("orszag-tang", "mesh.nx=24;mesh.ny=16;hydro.riemannSolver=hll", 4),
    ("orszag-tang", "mesh.nx=16;mesh.ny=24;hydro.riemannSolver=llf", 4),
    ("orszag-tang", "mesh.nx=16;mesh.ny=16;hydro.slope_type=1.0", 4),
    ("orszag-tang", "mesh.nx=16;mesh.ny=16;hydro.traceVersion=0", 3),          # slope_type forced to 0
    ("orszag-tang", "mesh.nx=3;mesh.ny=3", 3),
    ("orszag-tang", "mesh.nx=16;mesh.ny=16;hydro.riemannSolver=hllc", 2),       # MHD + hllc => zero hydro flux quirk
    ("mhd_BrioWu", "mesh.nx=20;mesh.ny=12;BrioWu.direction=3", 5),
    # 3D MHD plain: boundary types, solvers, isothermal
    ("mhd_BrioWu", "mesh.nx=12;mesh.ny=10;mesh.nz=8;BrioWu.direction=0;MHD.implementationVersion=4", 4),
    ("mhd_BrioWu", "mesh.nx=8;mesh.ny=12;mesh.nz=10;BrioWu.direction=2;MHD.implementationVersion=3;mesh.boundary_zmin=1;mesh.boundary_zmax=1", 4),
    ("orszag-tang3d", "mesh.nx=10;mesh.ny=8;mesh.nz=6;hydro.riemannSolver=hll", 3),
    ("orszag-tang3d", "mesh.nx=8;mesh.ny=8;mesh.nz=8;hydro.cIso=0.9;hydro.slope_type=1.0", 3),
    ("orszag-tang3d", "mesh.nx=3;mesh.ny=3;mesh.nz=3", 2),
    # 3D MHD rotating: without shearing box (periodic x), other sizes
    ("orszag-tang3d", "mesh.nx=8;mesh.ny=12;mesh.nz=6;MHD.omega0=0.05", 4),
    ("mhd_mri_3d", "mesh.nx=6;mesh.ny=8;mesh.nz=10;MHD.omega0=0.01;MRI.amp=0.2", 6),
    # hydro: 3D jet (inflow after the z fill), 2D solvers / slopes
    ("jet2d_cpu", "mesh.nx=12;mesh.ny=12;mesh.nz=12;mesh.zmax=1;jet.ijet=3;jet.offsetJet=2", 6),
    ("implode3d", "mesh.nx=12;mesh.ny=8;mesh.nz=1;hydro.riemannSolver=hll;hydro.slope_type=2.0", 5),
    ("implode3d", "mesh.nx=10;mesh.ny=10;mesh.nz=10;hydro.traceVersion=0;hydro.riemannSolver=hllc", 4),
    ("implode3d", "mesh.nx=2;mesh.ny=2;mesh.nz=2", 3),
    # the other EMF solvers and the positivity-preserving slopes on further paths
    ("mhd_mri_3d", "mesh.nx=6;mesh.ny=8;mesh.nz=10;MHD.magRiemannSolver=hlla;MRI.amp=0.2", 5),     # rotating + shear terms of the EMF
    ("mhd_mri_3d", "mesh.nx=6;mesh.ny=8;mesh.nz=6;MHD.magRiemannSolver=llf", 4),
    ("mhd_BrioWu", "mesh.nx=16;mesh.ny=12;BrioWu.direction=1;MHD.magRiemannSolver=llf", 5),
    ("mhd_BrioWu", "mesh.nx=16;mesh.ny=12;BrioWu.direction=1;MHD.magRiemannSolver=hllf;hydro.slope_type=3.0", 5),
    ("orszag-tang3d", "mesh.nx=8;mesh.ny=6;mesh.nz=10;hydro.slope_type=3.0;MHD.magRiemannSolver=hllf;hydro.cIso=0.8", 3),
    ("mhd_BrioWu", "mesh.nx=10;mesh.ny=8;mesh.nz=8;BrioWu.direction=2;MHD.implementationVersion=3;hydro.slope_type=3.0", 4),
    # per-cell gravity field (Keplerian disk): other solvers, direction-wise update, viscosity, 3D
    ("Keplerian_disk2d", "mesh.nx=14;mesh.ny=10;hydro.riemannSolver=approx", 5),
    ("Keplerian_disk2d", "mesh.nx=10;mesh.ny=12;mesh.nz=6;hydro.riemannSolver=hll;hydro.unsplitVersion=2;hydro.nu=0.003", 4),
    ("Keplerian_disk2d", "mesh.nx=12;mesh.ny=12;gravity.static=no", 3),
    # driven turbulence (round-off agreement): other sizes / solvers, zero input rate, with viscosity
    ("turbulence_hydro", "mesh.nx=10;mesh.ny=8;mesh.nz=12;hydro.riemannSolver=hll", 4),
    ("turbulence_hydro", "mesh.nx=8;mesh.ny=8;mesh.nz=8;turbulence.edot=0.0;hydro.nu=0.001", 3),
    ("turbulence_mhd", "mesh.nx=8;mesh.ny=10;mesh.nz=8;hydro.slope_type=2.0;turbulence.beta=2.0;MHD.eta=1e-4", 4),
    # stratified MRI box: other solvers, unsmoothed gravity, floor
    ("mhd_mri_3d_stratified", "mesh.nx=6;mesh.ny=8;mesh.nz=16;hydro.slope_type=2.0;MRI.amp=0.4;MHD.magRiemannSolver=hllf;hydro.riemannSolver=hll", 4),
    ("mhd_mri_3d_stratified", "mesh.nx=8;mesh.ny=6;mesh.nz=12;hydro.slope_type=1.0;MRI.amp=0.4;MRI.smoothGravity=no;MRI.floor=yes;mesh.zmin=-1.5;mesh.zmax=1.5", 4),
    # 2D branch of the rotating-frame step: other solvers / slopes / boundaries, implementationVersion 0 + gravity knobs ignored
    ("orszag-tang", "mesh.nx=16;mesh.ny=20;MHD.omega0=0.4;hydro.slope_type=3.0;MHD.magRiemannSolver=hlla", 5),
    ("mhd_BrioWu", "mesh.nx=20;mesh.ny=12;BrioWu.direction=1;MHD.omega0=0.25;hydro.riemannSolver=llf;MHD.magRiemannSolver=llf;mesh.boundary_ymin=1;mesh.boundary_ymax=1", 5),
    ("mhd_inertialWave_2d", "mesh.nx=8;mesh.ny=6;MHD.implementationVersion=0", 6),
]

RANDOM_STEPS = [
    ("orszag-tang", "mesh.nx=20;mesh.ny=14", 1.5),
    ("orszag-tang", "mesh.nx=14;mesh.ny=20;hydro.riemannSolver=hll", 3.0),
    ("mhd_BrioWu", "mesh.nx=16;mesh.ny=16", 2.5),
    ("orszag-tang", "mesh.nx=18;mesh.ny=14;MHD.omega0=0.6;hydro.cIso=0.7", 2.0),
    ("orszag-tang3d", "mesh.nx=10;mesh.ny=8;mesh.nz=6", 1.5),
    ("orszag-tang3d", "mesh.nx=6;mesh.ny=10;mesh.nz=8;hydro.riemannSolver=llf", 3.0),
    ("mhd_mri_3d", "mesh.nx=8;mesh.ny=10;mesh.nz=6;hydro.cIso=0.8;MHD.omega0=0.3", 1.0),
    ("mhd_mri_3d_stratified", "mesh.nx=8;mesh.ny=10;mesh.nz=8;hydro.slope_type=2.0;hydro.cIso=0.8;MHD.omega0=0.3", 1.0),
    ("mhd_mri_3d", "mesh.nx=8;mesh.ny=10;mesh.nz=6;hydro.cIso=0.8;MHD.omega0=0.3;mesh.boundary_xmin=3;mesh.boundary_xmax=3", 1.0),
    ("implode3d", "mesh.nx=10;mesh.ny=8;mesh.nz=6;hydro.riemannSolver=hllc", 3.0),
    ("implode3d", "mesh.nx=10;mesh.ny=8;mesh.nz=6", 3.0),
    ("implode3d", "mesh.nx=10;mesh.ny=8;mesh.nz=6;hydro.riemannSolver=hll;hydro.slope_type=2.0", 3.0),
    ("implode3d", "mesh.nx=12;mesh.ny=10;mesh.nz=1;hydro.riemannSolver=hllc", 3.0),
    ("jet2d_cpu", "mesh.nx=12;mesh.ny=16;jet.ijet=3;jet.offsetJet=2", 2.0),
]

BOUNDARY_CASES = [
    ("orszag-tang", "mesh.nx=8;mesh.ny=6"), ("mhd_BrioWu", "mesh.nx=6;mesh.ny=8"), ("jet2d_cpu", "mesh.nx=10;mesh.ny=8;jet.ijet=3;jet.offsetJet=2"),
    ("implode3d", "mesh.nx=6;mesh.ny=5;mesh.nz=4"), ("orszag-tang3d", "mesh.nx=6;mesh.ny=5;mesh.nz=4"),
    ("mhd_mri_3d", "mesh.nx=6;mesh.ny=8;mesh.nz=4;MHD.omega0=0.4"),
    ("mhd_BrioWu", "mesh.nx=4;mesh.ny=5;mesh.nz=6;mesh.boundary_xmin=1;mesh.boundary_ymax=1;mesh.boundary_zmin=3;mesh.boundary_zmax=3"),
]


# The bench's launch geometry: planes of > 32768 cells, where the XCD-aware workgroup order splits each XCD's y band
# into several sub-bands (rg_backend.h: rg_launch_planes, nsub > 1), with >= 2 chunks of the two-stream sweep and
# the LDS-tiled kernels' full-width tile rows.  Few planes keep the oracle at seconds per step.
BENCH_GEOMETRY = [
    ("mhd_mri_3d", "mesh.nx=512;mesh.ny=512;mesh.nz=16", 2),
    # the x-y cross-section of BASELINE config 5 (512 x 1024 x 512 over 8 GPUs): 33 x 129 tiles of the MHD sweep
    ("mhd_mri_3d", "mesh.nx=512;mesh.ny=1024;mesh.nz=16", 2),
    ("orszag-tang3d", "mesh.nx=256;mesh.ny=256;mesh.nz=24", 2),
    ("implode3d", "mesh.nx=512;mesh.ny=512;mesh.nz=8;hydro.riemannSolver=hllc", 2),
]


def check_core_plane_pieces(lib, base, ov):
    """rgpu_step_core_planes over a partition of [0,ksize) (odd cuts, out of order) == rgpu_step_core; so is the split form
    (rgpu_step_core_planes_split: the fluxes of the whole box once, then the update piece by piece -- the slab driver's order)"""
    p = lib.params_from_ini(ini(base), ov)
    U0 = lib.init_condition(ini(base), ov, p)
    ks = p.nz + 2 * p.ghostWidth
    cuts = [0, 1, 7, 8, ks - 15, ks]
    pieces = list(reversed(list(zip(cuts[:-1], cuts[1:]))))
    outs = []
    for mode in ("whole", "pieces", "split"):
        sv = Solver(p, lib)
        sv.upload(U0)
        sv.make_all_boundaries(0, 0.0, 0.0)
        dt = sv.compute_dt(0)
        sv.step_pre(0, dt, 0.0)
        if mode == "whole":
            sv.step_core(0, dt, 0.0)
        elif mode == "pieces":
            for a, b in pieces:
                sv.step_core_planes(0, dt, 0.0, a, b)
        else:
            sv.step_core_planes_split(0, dt, 0.0, 0, ks, 1)
            for a, b in pieces:
                sv.step_core_planes_split(0, dt, 0.0, a, b, 2)
        sv.step_post_a(0, dt, 0.0)
        sv.step_post_b(0, dt, 0.0)
        outs.append(sv.getDataHost(1))
        sv.close()
    assert np.array_equal(outs[0], outs[1])
    assert np.array_equal(outs[0], outs[2])


HISTORY_CASES = [
    ("mhd_mri_3d", "mesh.nx=8;mesh.ny=16;mesh.nz=8;MRI.amp=0.2", 6),
    ("orszag-tang3d", "mesh.nx=12;mesh.ny=10;mesh.nz=8", 3),
    ("orszag-tang", "mesh.nx=24;mesh.ny=16", 4),                       # 2D: history_default's mass / divB
    ("mhd_BrioWu", "mesh.nx=16;mesh.ny=12;mesh.nz=8;BrioWu.direction=0;MHD.implementationVersion=4", 3),
]


def check_history(lib, oracle, base, ov, nsteps):
    """device-reduced history diagnostics (rgpu_history_mri) == the oracle's sequential restatement of
    MHDRunBase::history_mri to round-off: the device adds in a different (fixed) order, so every sum is compared with
    a tolerance of 1e-12 x the sum of the magnitudes of its terms"""
    p = lib.params_from_ini(ini(base), ov)
    sv = Solver(p, lib)
    sv.start(lib.init_condition(ini(base), ov, p), nsteps)
    U = sv.getDataHost()
    got = sv.history_mri()
    sv.close()
    ref = dict(zip(Solver.HISTORY_NAMES, oracle.history_mri(p, U)))
    I = interior(U, p)
    cells = I[0].size
    dtau = 1.0 / cells          # dx*dy*dz / box volume
    b = np.abs(I[5:8]).sum() * dtau + 1e-300
    b2 = (I[5:8] ** 2).sum() * dtau + 1e-300
    mom = np.abs(I[2] * I[3] / I[0]).sum() * dtau + 1e-300
    scale = {"mass": ref["mass"], "maxwell": b2, "reynolds": mom, "magp": b2, "mean_Bx": b, "mean_By": b, "mean_Bz": b,
             "divB": np.abs(I[5:8]).sum() * 6.0 / min(p.dx, p.dy) + 1e-300}
    for k in Solver.HISTORY_NAMES:
        assert abs(got[k] - ref[k]) <= 1e-12 * scale[k], (k, got[k], ref[k], scale[k])


TURB_HISTORY_CASES = [
    ("turbulence_mhd_ou", "mesh.nx=12;mesh.ny=10;mesh.nz=8;turbulence-Ornstein-Uhlenbeck.bx=0.01;turbulence-Ornstein-Uhlenbeck.by=0.02", 5),
    ("turbulence_mhd", "mesh.nx=12;mesh.ny=12;mesh.nz=12", 4),
]


def check_history_turbulence(lib, oracle, base, ov, nsteps):
    """device-reduced history_turbulence (rgpu_history_turbulence) == the oracle's sequential restatement of
    MHDRunBase::history_turbulence to round-off (fixed parallel summation order, device cos / sin): every column is compared
    with a tolerance of 1e-11 x the magnitude of the terms of its sum"""
    p = lib.params_from_ini(ini(base), ov)
    sv = Solver(p, lib)
    try:
        attach_gravity(lib, base, ov, p, sv=sv)
        sv.start(lib.init_condition(ini(base), ov, p), nsteps)
        U = sv.getDataHost()
        got = sv.history_turbulence()
    finally:
        sv.close()
    ref = oracle.history_turbulence(p, U)
    gw = p.ghostWidth
    I = U[:, gw:-gw, gw:-gw, gw:-gw]
    dTau = p.dx * p.dy * p.dz / (p.xMax - p.xMin) / (p.yMax - p.yMin) / (p.zMax - p.zMin)
    bmax, mmax = np.abs(U[5:8]).max(), np.abs(I[2:5]).max()
    n = I[0].size
    mag = {1: bmax / min(p.dx, p.dy, p.dz) * n, 4: mmax * bmax * n * dTau, 10: mmax * n * dTau, 11: mmax * n * dTau, 12: mmax * n * dTau,
           15: bmax * n * dTau, 16: bmax * n * dTau, 17: bmax * n * dTau, 7: bmax * n * dTau, 8: bmax * n * dTau, 9: bmax * n * dTau}
    for q, (a, b) in enumerate(zip(got, ref)):
        tol = 1e-11 * max(abs(b), mag.get(q, 0.0))
        assert abs(a - b) <= tol, "history_turbulence column %d: device %r oracle %r tol %g" % (q, a, b, tol)


# ---- gates at the BASELINE sizes, for the exact library (equal bits) and for the tolerance-grade one (relative L2 < 1e-12) ----
OT_VARS = ["density", "energy", "mx", "my", "mz", "bx", "by", "bz"]


def orszag_tang_gate(lib, oracle, exact):
    """data/orszag-tang.ini as shipped (512^2, nstepmax = 50) against the oracle: north_star's gate, relative L2 < 1e-12 per
    variable; exact=True additionally demands every double and every dt equal.  Returns {variable: relative L2}."""
    p = lib.params_from_ini(ini("orszag-tang"))
    assert (p.nx, p.ny) == (512, 512)
    U0 = lib.init_condition(ini("orszag-tang"), "", p)
    ref, dts_ref, _ = oracle.run(p, U0, 50)
    sv = Solver(p, lib)
    try:
        dts = sv.start(U0, 50)
        got, ref = interior(sv.getDataHost(), p), interior(ref, p)
    finally:
        sv.close()
    errs = {}
    for v, name in enumerate(OT_VARS):
        errs[name] = float(rel_l2(got[v], ref[v]))
        assert errs[name] < L2_TOLERANCE, "Orszag-Tang %s: relative L2 %.3e" % (name, errs[name])
    if exact:
        assert np.array_equal(got, ref) and np.array_equal(np.array(dts), dts_ref)
    else:
        assert np.abs(np.array(dts) / dts_ref - 1.0).max() < 1e-11
    return errs


def divB_max(U, p):
    gw = p.ghostWidth
    bx, by, bz = U[5], U[6], U[7]
    s = (slice(gw, -gw),) * 3
    sx = (slice(gw, -gw), slice(gw, -gw), slice(gw + 1, -gw + 1 if gw > 1 else None))
    sy = (slice(gw, -gw), slice(gw + 1, -gw + 1 if gw > 1 else None), slice(gw, -gw))
    sz = (slice(gw + 1, -gw + 1 if gw > 1 else None), slice(gw, -gw), slice(gw, -gw))
    d = (bx[sx] - bx[s]) / p.dx + (by[sy] - by[s]) / p.dy + (bz[sz] - bz[s]) / p.dz
    return float(np.abs(d).max())


def mri_headline_size_properties(lib):
    """512^3 MRI (the bench workload): div B at round-off, mass conserved to round-off, finite fields"""
    ov = "mesh.nx=512;mesh.ny=512;mesh.nz=512"
    p = lib.params_from_ini(ini("mhd_mri_3d"), ov)
    U0 = lib.init_condition(ini("mhd_mri_3d"), ov, p)
    sv = Solver(p, lib)
    try:
        sv.start(U0, 0)
        A = sv.getDataHost()
        m0 = interior(A, p)[0].sum(dtype=np.longdouble)
        b_scale = float(np.abs(A[7]).max()) / p.dx
        d0 = divB_max(A, p)
        del U0, A
        for _ in range(3):
            sv.oneStepIntegration()
        B = sv.getDataHost()
    finally:
        sv.close()
    assert np.isfinite(B).all()
    m1 = interior(B, p)[0].sum(dtype=np.longdouble)
    assert abs(float((m1 - m0) / m0)) < 1e-13
    assert divB_max(B, p) <= max(d0, 1e-13 * b_scale) + 1e-12 * b_scale


def implode_bench_size_properties(lib):
    """256^3 implode, HLLC, reflecting walls: mass and energy conserved to round-off, x <-> y symmetry of the solution"""
    ov = "mesh.nx=256;mesh.ny=256;mesh.nz=256;hydro.riemannSolver=hllc"
    p = lib.params_from_ini(ini("implode3d"), ov)
    U0 = lib.init_condition(ini("implode3d"), ov, p)
    sv = Solver(p, lib)
    try:
        sv.start(U0, 5)
        A = interior(sv.getDataHost(), p)
    finally:
        sv.close()
    I0 = interior(U0, p)
    for v in (0, 1):
        a, b = A[v].sum(dtype=np.longdouble), I0[v].sum(dtype=np.longdouble)
        assert abs(float((a - b) / b)) < 1e-13
    # x <-> y transposition symmetry: density invariant, mx <-> my
    assert np.allclose(A[0], A[0].transpose(0, 2, 1), rtol=0, atol=1e-12)
    assert np.allclose(A[2], A[3].transpose(0, 2, 1), rtol=0, atol=1e-12)


def long_run_error_growth(lib, oracle, base, ov, nsteps, every):
    """relative L2 (all variables together) of `lib` against the oracle after every `every` steps of one run -- how round-off
    differences of the tolerance-grade arithmetic grow on a long run.  Returns [(step, relative L2, max |dt / dt_ref - 1|)]."""
    p = lib.params_from_ini(ini(base), ov)
    U0 = lib.init_condition(ini(base), ov, p)
    out = []
    sv = Solver(p, lib)
    try:
        sv.start(U0, 0)      # the init part of start(): ghost fill of the initial state, both arrays
        dts = []
        for n in range(1, nsteps + 1):
            dts.append(sv.oneStepIntegration())
            if n % every == 0 or n == nsteps:
                ref, dts_ref, _ = oracle.run(p, U0, n)
                got = interior(sv.getDataHost(), p)
                out.append((n, float(rel_l2(got, interior(ref, p))), float(np.abs(np.array(dts) / np.asarray(dts_ref)[:n] - 1.0).max())))
    finally:
        sv.close()
    return out


def check_public_ghost_fill_invalidates_fused_dt(lib, oracle, base, ov):
    """ADVICE (round 2): the step's last kernel leaves the CFL maximum of the new state in a device slot and the next compute_dt
    only reads it back.  A ghost fill called from OUTSIDE the step with a non-periodic MHD face overwrites the field the CT update
    left on the first high ghost face, which compute_dt_mhd reads: the slot is then stale.  After rgpu_make_all_boundaries the
    library must scan again -- the value must equal the oracle's compute_dt of the downloaded state -- and rgpu_invalidate_dt must
    do the same for a host program that wrote into the arrays itself."""
    p = lib.params_from_ini(ini(base), ov)
    U0 = lib.init_condition(ini(base), ov, p)
    sv = Solver(p, lib)
    try:
        sv.start(U0, 3)                               # three steps: the slot holds the scan of the state after step 3
        par = sv.nStep % 2
        sv.make_all_boundaries(par, sv.totalTime, 0.0)   # public ghost fill of the CURRENT state
        got = sv.compute_inv_dt(par)
        ref = oracle.compute_inv_dt(p, sv.getDataHost(par))
        assert got == ref, (got, ref)
        # rgpu_invalidate_dt (a host program that wrote into adopted arrays calls it): the next compute_dt scans again
        sv.oneStepIntegration()
        par = sv.nStep % 2
        ref = oracle.compute_inv_dt(p, sv.getDataHost(par))
        sv.enable_timers(True)
        sv.reset_timers()
        fused = sv.compute_inv_dt(par)                 # read back from the slot the update kernel filled: no scan kernel
        t_fused = sv.timers().get("dt", 0.0)
        assert lib.lib.rgpu_invalidate_dt(sv.ctx) == 0
        rescanned = sv.compute_inv_dt(par)
        t_scan = sv.timers().get("dt", 0.0)
        sv.enable_timers(False)
        assert fused == ref and rescanned == ref, (fused, rescanned, ref)
        if "emulation" not in lib.backend:             # (the emulation backend has no phase timers)
            assert t_fused == 0.0 and t_scan > 0.0, (t_fused, t_scan)
    finally:
        sv.close()


# ---- direct differential test of the Alfven selection in the 2D HLLD edge solver (rgpu_selftest_alfven) -------------------------
def alfven_samples(n, seed):
    """(kept for the last few (n, seed): the four parameter sets of the selection test run on the same samples; read-only)"""
    key = (int(n), int(seed))
    if key not in _ALFVEN_SAMPLES:
        while len(_ALFVEN_SAMPLES) >= 3:
            _ALFVEN_SAMPLES.pop(next(iter(_ALFVEN_SAMPLES)))
        S, kind = _alfven_samples(n, seed)
        S.flags.writeable = False
        kind.flags.writeable = False
        _ALFVEN_SAMPLES[key] = (S, kind)
    return _ALFVEN_SAMPLES[key]


_ALFVEN_SAMPLES = {}


def _alfven_samples(n, seed):
    """n edge problems (SoA [36, n]): half random rough states, half adversarial -- built to sit on the selection's margins:
      * dv / dS = 1 +- k 2^-52 (k <= 2^14): a uniform flow (the star ratio t = dv / dS is exactly 1) whose velocities are nudged
        by k ulp, so that candidates and their star partners tie, nearly tie, or differ by a hair more than the 2^-40 margin;
      * equal candidates (all four states identical, or pairwise identical);
      * |b| from 1e-120 to 1e3, density ratios 2^+-60 between the states;
      * the kinds interleaved lane by lane, so that a wave of 64 samples holds both routes."""
    rng = np.random.RandomState(seed)
    S = np.empty((36, n))
    kind = rng.randint(0, 8, n)                    # per LANE: neighbours in a wave differ
    # base: random states
    for q in range(4):
        S[8 * q + 0] = np.exp(rng.uniform(-2, 2, n))            # r
        S[8 * q + 1] = np.exp(rng.uniform(-3, 2, n))            # p
        S[8 * q + 2:8 * q + 5] = rng.normal(0, 1.5, (3, n))     # u v w
        S[8 * q + 5:8 * q + 8] = rng.normal(0, 1.0, (3, n))     # a b c
    uni = kind >= 2                                # adversarial kinds start from a uniform flow: copy state 0 into the others
    for q in range(1, 4):
        S[8 * q:8 * q + 8, uni] = S[0:8, uni]
    ulp = 2.0 ** -52
    k = np.floor(np.exp2(rng.uniform(0, 14, n))) * np.where(rng.rand(n) < 0.5, -1.0, 1.0)
    # kind 2: exactly uniform (every pair ties).  3: one velocity of one state nudged by k ulp.  4: all velocities of two states nudged
    m = kind == 3
    q3 = rng.randint(0, 4, n); c3 = rng.randint(2, 4, n)
    for q in range(4):
        for c in (2, 3):
            sel = m & (q3 == q) & (c3 == c)
            S[8 * q + c, sel] *= 1.0 + k[sel] * ulp
    m = kind == 4
    for q in (1, 2):
        for c in (2, 3):
            S[8 * q + c, m] *= 1.0 + k[m] * ulp * (1 if q == 1 else -1)
    # kind 5: pairwise identical states (LL == LR, RL == RR) with a k-ulp density nudge across the pair
    m = kind == 5
    S[8:16, m] = S[0:8, m] * 1.0
    S[8:9, m] *= 1.0 + k[m] * ulp
    S[16:24, m] = S[0:8, m]
    S[24:32, m] = S[8:16, m]
    # kind 6: field scaled over 1e-120 .. 1e3 (all states alike, so that candidates stay comparable), velocities nudged
    m = kind == 6
    scale = np.exp(rng.uniform(np.log(1e-120), np.log(1e3), n))
    for q in range(4):
        S[8 * q + 5:8 * q + 8, m] *= scale[m]
        S[8 * q + 2, m] *= 1.0 + (q - 1.5) * k[m] * ulp
    # kind 7: density ratios 2^+-60 between the states (pressure follows, so that sound speeds stay finite), random fields
    m = kind == 7
    for q in range(1, 4):
        e = rng.randint(-60, 61, n).astype(float)
        S[8 * q + 0, m] = S[0, m] * np.exp2(e[m])
        S[8 * q + 1, m] = S[1, m] * np.exp2(e[m])
        S[8 * q + 5:8 * q + 8, m] = rng.normal(0, 1.0, (3, int(m.sum())))
    # kind 1 (random): field scale over a wide range too
    m = kind == 1
    scale = np.exp(rng.uniform(np.log(1e-120), np.log(1e3), n))
    for q in range(4):
        S[8 * q + 5:8 * q + 8, m] *= scale[m]
    # electric fields of the corner states, E = u b - v a (riemann_mhd.h:1115-1140)
    for q in range(4):
        S[32 + q] = S[8 * q + 2] * S[8 * q + 6] - S[8 * q + 3] * S[8 * q + 5]
    return np.ascontiguousarray(S), kind


def check_alfven_selftest(lib, n, seed, smallc, iso=False):
    """e_select == e_reference, bit for bit; returns (waves on the selection route, waves on the reference route)"""
    import ctypes as C
    from conftest import ini
    p = lib.params_from_ini(ini("mhd_mri_3d" if iso else "orszag-tang3d"), "mesh.nx=8;mesh.ny=8;mesh.nz=8")
    p.smallc = smallc
    S, kind = alfven_samples(n, seed)
    e_sel, e_ref = np.empty(n), np.empty(n)
    route = np.empty(n, dtype=np.int32)
    f = lib.lib.rgpu_selftest_alfven
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    P = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    rc = f(C.byref(p), n, P(S), P(e_sel), P(e_ref), route.ctypes.data_as(C.POINTER(C.c_int)))
    assert rc == 0, rc
    a, b = e_sel.view(np.uint64), e_ref.view(np.uint64)
    bad = (a != b) & ~(np.isnan(e_sel) & np.isnan(e_ref))
    assert not bad.any(), "%d of %d edge problems differ between the selection and the reference's sequence; first: sample %d kind %d: %r vs %r" % (
        int(bad.sum()), n, int(np.argmax(bad)), int(kind[np.argmax(bad)]), e_sel[np.argmax(bad)], e_ref[np.argmax(bad)])
    assert np.isfinite(e_ref).mean() > 0.9, "the samples are mostly not finite: %r" % np.isfinite(e_ref).mean()
    return int((route == 0).sum()), int((route == 1).sum()), kind, route


def check_fused_fill(lib, oracle, base, ov, seed=5, t0=3.7, dt=0.9):
    """rgpu_step_fill_planes_pair (one launch for x and y faces / the shearing-box remap, corners included, two plane ranges) against the
    oracle's separate passes -- X then Y, or the whole shearing-box sequence Y, shear, Z, Y -- on a random state: every ghost cell of the
    planes asked for equal, every other plane untouched"""
    p = lib.params_from_ini(ini(base), ov)
    U = random_state(p, seed)
    gw, ks = p.ghostWidth, p.nz + 2 * p.ghostWidth
    ref = U.copy()
    shear = bool(p.shearingBoxEnabled)
    if shear:
        oracle.make_all_boundaries(p, ref, t0, dt)
    else:
        oracle.make_boundaries(p, ref, 1)
        oracle.make_boundaries(p, ref, 2)
    r1, r2 = (gw, 2 * gw), (p.nz, p.nz + gw)            # the planes a slab sends
    if r2[0] < r1[1]:
        r1, r2 = (gw, p.nz + gw), (0, 0)
    sv = Solver(p, lib)
    try:
        sv.upload(U, both=True)
        sv.step_fill_planes_pair(1, dt, t0, r1, r2)      # step 1 writes U[0]
        got = sv.getDataHost(0)
        planes = list(range(*r1)) + list(range(*r2))
        for k in range(ks):
            if k in planes:
                assert np.array_equal(got[:, k], ref[:, k]), "%s: plane %d differs from the separate passes in %d doubles" % (base, k, int((got[:, k] != ref[:, k]).sum()))
            else:
                assert np.array_equal(got[:, k], U[:, k]), "%s: plane %d outside the ranges was touched" % (base, k)
        sv.upload(U, both=True)                          # one range, all interior planes
        sv.step_fill_planes_pair(1, dt, t0, (gw, ks - gw), (0, 0))
        got = sv.getDataHost(0)
        assert np.array_equal(got[:, gw:ks - gw], ref[:, gw:ks - gw])
    finally:
        sv.close()
