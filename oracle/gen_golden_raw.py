#!/usr/bin/env python3
"""Xsmurf (.xsm), NRRD (.nrrd) and text-mode VTI (.vti, outputVtkAscii) files written by the reference binary (oracle/_ref/euler_cpu; HydroRunBase.cpp:2520-2562,
4266-4335) for two tiny runs -> tests/golden/raw/<case>/: byte-for-byte fixtures of the run driver's two raw output formats.
usage: python oracle/gen_golden_raw.py"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from gen_golden import apply_overrides  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "raw")
CASES = {
    "ot2d_8x6": ("orszag-tang", "mesh.nx=8;mesh.ny=6", 2),
    "implode3d_6x4x5": ("implode3d", "mesh.nx=6;mesh.ny=4;mesh.nz=5;hydro.riemannSolver=hllc", 2),
}


def main():
    listing = {}
    for name, (base, ov, last) in sorted(CASES.items()):
        full = ov + ";run.nstepmax=%d;run.noutput=%d;run.tend=1e9;output.outputVtk=yes;output.outputVtkAscii=yes;output.outputHdf5=no;output.outputXsm=yes;output.outputNrrd=yes;output.outputDir=./" % (last, last)
        ini = apply_overrides(open(os.path.join(ROOT, "configs", base + ".ini")).read(), full)
        dst = os.path.join(OUT, name)
        shutil.rmtree(dst, ignore_errors=True)
        os.makedirs(dst)
        with tempfile.TemporaryDirectory() as td:
            open(os.path.join(td, "case.ini"), "w").write(ini)
            subprocess.run([os.path.join(HERE, "_ref", "euler_cpu"), "--param", "case.ini"], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, check=True)
            for f in sorted(os.listdir(td)):
                if f.endswith(("%07d.xsm" % last, "%07d.nrrd" % last, "%07d.vti" % last)):
                    shutil.copy(os.path.join(td, f), dst)
            # the default, appended-raw .vti of the same run
            open(os.path.join(td, "bin.ini"), "w").write(apply_overrides(ini, "output.outputVtkAscii=no;output.outputXsm=no;output.outputNrrd=no"))
            bd = os.path.join(td, "bin"); os.makedirs(bd)
            subprocess.run([os.path.join(HERE, "_ref", "euler_cpu"), "--param", "../bin.ini"], cwd=bd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, check=True)
            for f in sorted(os.listdir(bd)):
                if f.endswith("%07d.vti" % last):
                    shutil.copy(os.path.join(bd, f), os.path.join(dst, f.replace(".vti", ".binary.vti")))
        listing[name] = {"base": base, "overrides": ov, "last_step": last}
        print(name, sorted(os.listdir(dst)))
    json.dump(listing, open(os.path.join(OUT, "cases.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
