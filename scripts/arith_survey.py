#!/usr/bin/env python3
"""GPU: relative L2 distance of every golden fixture (outputs of the reference binary) from the run of a library variant
(RGPU_LIB=<path of a contracted-arithmetic build>), one line per case and step count."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden_cases, ini, load_golden
from parity_checks import attach_gravity, rel_l2
from ramsesgpu_amd.solver import Library, Solver, interior, lib_path

lib = Library(lib_path())
worst = 0.0
for name, case in sorted(golden_cases().items()):
    try:
        p = lib.params_from_ini(ini(case["base"]), case["overrides"])
        g = load_golden(name)
        out = []
        for s in case["steps"]:
            U0 = lib.init_condition(ini(case["base"]), case["overrides"], p)
            sv = Solver(p, lib)
            attach_gravity(lib, case["base"], case["overrides"], p, sv=sv)
            sv.start(U0, s)
            got = interior(sv.getDataHost(), p)
            e = rel_l2(got, g["step_%d" % s]); worst = max(worst, e)
            out.append("%d:%.2e(%d)" % (s, e, int((got != g["step_%d" % s]).sum())))
            sv.close()
        print("%-36s %s" % (name, "  ".join(out)), flush=True)
    except Exception as e:  # noqa: BLE001
        print("%-36s ERROR %r" % (name, e), flush=True)
print("worst relative L2: %.3e" % worst)
