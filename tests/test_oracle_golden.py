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
