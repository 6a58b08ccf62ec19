"""The oracle (CPU restatement) against fixtures produced by THE REFERENCE ITSELF (oracle/gen_golden.py ran
oracle/_ref/euler_cpu built from /root/reference).  Bar: bit-identical doubles, dt sequence included.
This is what pins the oracle; the HIP path is then checked against the oracle and the same fixtures."""
import numpy as np
import pytest

from conftest import golden_cases, ini, load_golden
from ramsesgpu_amd.solver import interior

CASES = sorted(golden_cases().items())


@pytest.mark.parametrize("name,case", CASES, ids=[c[0] for c in CASES])
def test_oracle_reproduces_reference_bit_for_bit(name, case, oracle, product_lib):
    L = product_lib  # host entry points only: parameter file -> params, initial condition
    p = L.params_from_ini(ini(case["base"]), case["overrides"])
    g = load_golden(name)
    for s in case["steps"]:
        U0 = L.init_condition(ini(case["base"]), case["overrides"], p)
        oracle.set_gravity_field(L.init_gravity(ini(case["base"]), case["overrides"], p))   # h_gravity, where the problem has one
        oracle.set_forcing_field(L.init_forcing(ini(case["base"]), case["overrides"], p))   # h_randomForcing, likewise
        U, dts, t = oracle.run(p, U0, s)
        ref = g["step_%d" % s]
        got = interior(U, p)
        assert got.shape == ref.shape
        assert np.array_equal(got, ref), "%s step %d: %d doubles differ from the reference" % (name, s, (got != ref).sum())
        # dt log of the reference (line N carries the dt of step N-1): the MHD run class prints 12 decimals
        # (MHDRunGodunov.cpp:3955-3958), the hydro one 8
        if s >= 2 and len(g["log_dt"]) > s - 1:
            digits = 0.6e-12 if p.mhdEnabled else 0.6e-8
            np.testing.assert_allclose(dts[: s - 1], g["log_dt"][1:s], rtol=0, atol=digits * max(1.0, dts.max()))
        if s == max(case["steps"]) and np.isfinite(g["total_time"]):
            assert abs(t - float(g["total_time"])) <= 1e-11 * max(1.0, abs(t))


@pytest.mark.parametrize("name", ["mri_8x16x8_history", "ot3d_12_history"])
def test_oracle_history_matches_reference_history_file(name, oracle, product_lib):
    """the reference writes <prefix>_history.txt with 6 significant digits (MHDRunBase.cpp:3596-3602, 3401-3402): one
    row before every step.  The oracle's restatement of history_mri / history_default on its own (bit-identical) states
    must print the same digits."""
    from conftest import golden_cases
    case = golden_cases()[name]
    L = product_lib
    p = L.params_from_ini(ini(case["base"]), case["overrides"])
    g = load_golden(name)
    H = g["history"]
    for n in range(H.shape[0]):
        U0 = L.init_condition(ini(case["base"]), case["overrides"], p)
        U, dts, t = oracle.run(p, U0, n)
        h = oracle.history_mri(p, U)   # mass, maxwell, reynolds, magp, mean_Bx, mean_By, mean_Bz, divB
        if H.shape[1] == 11:           # totalTime dt mass maxwell reynolds maxwell+reynolds magp mean_B(3) divB
            mine = [h[0], h[1], h[2], h[1] + h[2], h[3], h[4], h[5], h[6], h[7]]
        else:                          # history_default: totalTime dt mass divB
            mine = [h[0], h[7]]
        for a, b in zip(mine, H[n, 2:]):
            assert float("%.5e" % a) == float("%.5e" % b) or abs(a - b) <= 2e-6 * abs(b), (n, a, b)


def test_oracle_turbulence_history_matches_reference_history_file(oracle, product_lib):
    """history_turbulence (MHDRunBase.cpp:3626-3810) of the Ornstein-Uhlenbeck MHD run: the reference prints a row whenever
    the time crosses the next multiple of dtHist (MHDRunGodunov.cpp:3975-3984); the oracle's restatement on its own
    (bit-identical) states prints the same 6 significant digits.  Columns that are pure cancellation noise (divB, the
    momenta means, the DFT amplitudes of an almost uniform field) are compared against the scale of their terms."""
    from conftest import golden_cases
    name = "turb_ou_mhd_12_history"
    case = golden_cases()[name]
    L = product_lib
    p = L.params_from_ini(ini(case["base"]), case["overrides"])
    H = load_golden(name)["history"]
    assert H.shape[1] == 20 and H.shape[0] >= 4
    U0 = L.init_condition(ini(case["base"]), case["overrides"], p)
    _, dts, _ = oracle.run(p, U0.copy(), 8)
    times = np.concatenate([[0.0], np.cumsum(dts)])
    for row in H:
        n = int(np.argmin(np.abs(times - row[0])))
        assert abs(times[n] - row[0]) <= 1e-5 * max(1.0, row[0])
        U, _, _ = oracle.run(p, L.init_condition(ini(case["base"]), case["overrides"], p), n)
        h = oracle.history_turbulence(p, U)
        scale = {1: 1e-4 / p.dx, 10: 1.0, 11: 1.0, 12: 1.0, 15: 1e-8, 16: 1e-8, 17: 1e-8, 4: 1e-4}   # magnitudes of the terms of the cancelling sums
        for q, (a, b) in enumerate(zip(h, row[2:])):
            if q in scale:
                assert abs(a - b) <= 1e-12 * scale[q] * p.nx * p.ny * p.nz + 6e-6 * abs(b), (n, q, a, b)   # 6 printed digits
            else:
                assert float("%.5e" % a) == float("%.5e" % b) or abs(a - b) <= 2e-6 * abs(b), (n, q, a, b)


@pytest.mark.parametrize("base,ov,nsteps,nthreads", [
    ("mhd_mri_3d", "mesh.nx=16;mesh.ny=32;mesh.nz=16", 6, 3),                  # rotating + shearing box (the headline workload)
    ("mhd_mri_3d", "mesh.nx=8;mesh.ny=12;mesh.nz=10;MHD.omega0=0.02", 4, 16),  # more threads than planes per slab
    ("orszag-tang3d", "mesh.nx=12;mesh.ny=10;mesh.nz=14", 4, 4),               # plain 3D MHD
])
def test_threaded_step_equals_sequential(base, ov, nsteps, nthreads, oracle, product_lib):
    """the all-cores CPU baseline (orc_run_mt: z-slab threads, fluxes stored, gather update in scatter order) reproduces the
    sequential restatement -- which the golden fixtures pin to the reference binary -- bit for bit, dt sequence included"""
    p = product_lib.params_from_ini(ini(base), ov)
    U0 = product_lib.init_condition(ini(base), ov, p)
    a, da, ta = oracle.run(p, U0, nsteps)
    b, db, tb = oracle.run_mt(p, U0, nsteps, nthreads)
    assert np.array_equal(da, db) and ta == tb
    assert np.array_equal(interior(a, p), interior(b, p))


@pytest.mark.parametrize("mode", [1, 2])
def test_pinned_threads_change_nothing_but_the_placement(mode, oracle, product_lib):
    """cpu_baseline_all_cores pins its threads (placement 1: all allowed CPUs in NUMA-node order, 2: the first NUMA node alone) and
    lets every z-slab be first touched by its thread: same doubles, same time steps as the unpinned and the sequential runs;
    the pinning is undone for the calling thread's process afterwards (threads are pinned one by one, never the process)"""
    import os
    base, ov, nsteps = "mhd_mri_3d", "mesh.nx=8;mesh.ny=12;mesh.nz=16;MRI.amp=0.2", 4
    p = product_lib.params_from_ini(ini(base), ov)
    U0 = product_lib.init_condition(ini(base), ov, p)
    a, da, ta = oracle.run(p, U0, nsteps)
    before = os.sched_getaffinity(0)
    ncpu = oracle.set_thread_placement(mode)
    try:
        assert 1 <= ncpu <= len(before)
        secs, used = oracle.run_mt_scan(p, U0, nsteps, [4, 2], 3)
        b, db, tb = oracle.run_mt(p, U0, nsteps, 5)
    finally:
        assert oracle.set_thread_placement(0) == 0
    assert len(secs) == nsteps and used in (4, 2)
    assert os.sched_getaffinity(0) == before
    assert np.array_equal(da, db) and ta == tb and np.array_equal(interior(a, p), interior(b, p))
