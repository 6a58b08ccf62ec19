#!/usr/bin/env python3
"""HDF5 fixtures written by THE REFERENCE ITSELF, built with -DUSE_HDF5 against the HDF5 1.10 of the image
(oracle/Makefile.ref, target h5 -> oracle/_ref/euler_cpu_h5): for every case the files <prefix>_NNNNNNN.h5 of two output
steps and the .xmf index go to tests/golden/h5/<case>/ (cases listed in tests/golden/h5/cases.json).  Test infrastructure:
the writer of the run driver must produce the same datasets, its reader must resume from the reference's file.

usage: python oracle/gen_golden_h5.py
"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from gen_golden import apply_overrides  # noqa: E402

REF_BIN = os.path.join(HERE, "_ref", "euler_cpu_h5")
OUT = os.path.join(ROOT, "tests", "golden", "h5")

# name -> (base config, overrides, ghostIncluded, step of the first kept file, last step)
CASES = {
    "ot2d_16x12": ("orszag-tang", "mesh.nx=16;mesh.ny=12", "no", 3, 6),
    "ot2d_16x12_ghosts": ("orszag-tang", "mesh.nx=16;mesh.ny=12", "yes", 3, 6),
    "mri_4x8x4_ghosts": ("mhd_mri_3d", "mesh.nx=4;mesh.ny=8;mesh.nz=4;MRI.amp=0.2", "yes", 3, 6),   # shearing box: restart needs the ghosts
    "implode3d_6_hllc": ("implode3d", "mesh.nx=6;mesh.ny=6;mesh.nz=6;hydro.riemannSolver=hllc", "no", 2, 4),
    "jet2d_8x20": ("jet2d_cpu", "mesh.nx=8;mesh.ny=20;jet.ijet=2;jet.offsetJet=1", "no", 2, 4),
}


def main():
    subprocess.check_call(["make", "-C", HERE, "-f", "Makefile.ref", "h5"])
    os.makedirs(OUT, exist_ok=True)
    listing = {}
    for name, (base, ov, ghosts, s0, s1) in sorted(CASES.items()):
        full = ov + ";run.nstepmax=%d;run.noutput=%d;run.tend=1e9;run.nlog=1;output.outputVtk=no;output.outputHdf5=yes;output.ghostIncluded=%s;output.outputDir=./" % (s1, s0, ghosts)
        ini = apply_overrides(open(os.path.join(ROOT, "configs", base + ".ini")).read(), full)
        prefix = re.search(r"outputPrefix=(\S+)", ini).group(1)
        dst = os.path.join(OUT, name)
        shutil.rmtree(dst, ignore_errors=True)
        os.makedirs(dst)
        with tempfile.TemporaryDirectory() as td:
            open(os.path.join(td, "case.ini"), "w").write(ini)
            subprocess.run([REF_BIN, "--param", "case.ini"], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, check=True)
            for s in (s0, s1):
                shutil.copy(os.path.join(td, "%s_%07d.h5" % (prefix, s)), dst)
            shutil.copy(os.path.join(td, prefix + ".xmf"), dst)
        listing[name] = {"base": base, "overrides": ov, "ghostIncluded": ghosts, "prefix": prefix, "restart_step": s0, "last_step": s1}
        print(name, sorted(os.listdir(dst)))
    json.dump(listing, open(os.path.join(OUT, "cases.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
