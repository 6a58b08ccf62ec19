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
