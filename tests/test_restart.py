"""[run] restart=yes through the run driver (rgpuh_run / euler_hip): a run resumed from the .vti an earlier run wrote at
step N -- interior fields as raw doubles, step count and time in the header, the Ornstein-Uhlenbeck process in a sidecar --
ends in exactly the state of the uninterrupted run.  (The reference restarts from HDF5, HydroRunBase.cpp:7033-7066,
4818-5160; this image has no HDF5 library, so its binary cannot produce a fixture: the property tested is the one that
defines a restart.)  CPU: TEST-ONLY emulation library; GPU: the product."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, ini

sys.path.insert(0, os.path.join(ROOT, "oracle"))

# (ini, overrides, resume from the ghost-inclusive raw dump instead of the .vti)
CASES = [
    ("orszag-tang", "mesh.nx=24;mesh.ny=16", False),
    ("orszag-tang", "mesh.nx=24;mesh.ny=16", True),
    ("mhd_mri_3d", "mesh.nx=8;mesh.ny=12;mesh.nz=8;MRI.amp=0.2", True),                         # shearing box: needs the ghosts
    ("implode3d", "mesh.nx=12;mesh.ny=12;mesh.nz=12;hydro.riemannSolver=hllc", False),
    ("jet2d_cpu", "mesh.nx=16;mesh.ny=40;jet.ijet=4;jet.offsetJet=3", False),
    ("turbulence_hydro_ou", "mesh.nx=10;mesh.ny=10;mesh.nz=10", False),                          # forcing process resumed from its sidecar
    ("turbulence_mhd_ou", "mesh.nx=8;mesh.ny=8;mesh.nz=8;history.enabled=no", True),
    ("rayleigh_taylor_gpu_3d_mhd", "mesh.nx=8;mesh.ny=8;mesh.nz=16", False),                     # static gravity rebuilt
]


def run(lib, base, ov, outdir, dump="no"):
    err = C.create_string_buffer(512)
    mc = C.c_double(0)
    full = ov + ";output.outputVtk=yes;output.outputHdf5=%s;output.ghostIncluded=yes;output.outputDir=%s" % (dump, outdir)
    n = lib.lib.rgpuh_run(ini(base).encode(), full.encode(), C.byref(mc), err, 512)
    assert n >= 0, err.value
    return n


def check_restart(lib, base, ov, tmp_path, from_dump):
    """from_dump: resume from the raw dump with ghost cells ([output] outputHdf5=yes, ghostIncluded=yes -> *.rgr) instead of
    the interior-only .vti -- what a shearing-box run needs (the field on the first high x face is part of the state)"""
    from gen_golden import read_vti
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir(); b.mkdir()
    dump = "yes" if from_dump else "no"
    assert run(lib, base, ov + ";run.nstepmax=10;run.noutput=5;run.tend=1e9", a, dump) == 10
    files = sorted(f for f in os.listdir(a) if f.endswith(".vti"))
    assert len(files) == 3 and files[1].endswith("0000005.vti"), files
    # resume in another directory from step 5
    for f in os.listdir(a):
        if "0000005" in f:
            (b / f).write_bytes((a / f).read_bytes())
    src = files[1].replace(".vti", ".rgr") if from_dump else files[1]
    assert run(lib, base, ov + ";run.nstepmax=10;run.noutput=5;run.tend=1e9;run.restart=yes;run.restart_filename=%s" % src, b, dump) == 10
    fa, _ = read_vti(str(a / files[2]))
    fb, _ = read_vti(str(b / files[2]))
    assert sorted(fa) == sorted(fb)
    for name in fa:
        assert np.array_equal(fa[name], fb[name]), "%s differs after the restart (%d values)" % (name, int((fa[name] != fb[name]).sum()))
    ha = open(a / files[2], "rb").read(300).split(b"\n")[1]
    hb = open(b / files[2], "rb").read(300).split(b"\n")[1]
    assert ha == hb and b"nStep=10" in ha       # same step count and time, to the last bit (hex float)


IDS = ["%s-%s" % (c[0], "dump" if c[2] else "vti") for c in CASES]


@pytest.mark.parametrize("base,ov,from_dump", CASES, ids=IDS)
def test_restart_equals_uninterrupted_run_emu(base, ov, from_dump, emu_lib, tmp_path):
    check_restart(emu_lib, base, ov, tmp_path, from_dump)


@pytest.mark.gpu
@pytest.mark.parametrize("base,ov,from_dump", CASES, ids=IDS)
def test_restart_equals_uninterrupted_run_gpu(base, ov, from_dump, gpu_lib, tmp_path):
    check_restart(gpu_lib, base, ov, tmp_path, from_dump)


def test_restart_rejects_a_file_of_another_box(emu_lib, tmp_path):
    run(emu_lib, "orszag-tang", "mesh.nx=16;mesh.ny=16;run.nstepmax=2;run.noutput=1", tmp_path)
    err = C.create_string_buffer(512)
    mc = C.c_double(0)
    f1 = [f for f in os.listdir(tmp_path) if f.endswith("0000001.vti")][0]
    ov = "mesh.nx=24;mesh.ny=16;run.restart=yes;run.restart_filename=%s;output.outputDir=%s" % (f1, tmp_path)
    n = emu_lib.lib.rgpuh_run(ini("orszag-tang").encode(), ov.encode(), C.byref(mc), err, 512)
    assert n < 0 and b"another box" in err.value
    ov = "run.restart=yes;run.restart_upscale=yes;run.restart_filename=x.vti"
    n = emu_lib.lib.rgpuh_run(ini("orszag-tang").encode(), ov.encode(), C.byref(mc), err, 512)
    assert n < 0 and b"restart_upscale" in err.value
