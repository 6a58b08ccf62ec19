"""[output] outputHdf5 / [run] restart from .h5 through the run driver, against files written by THE REFERENCE built with
-DUSE_HDF5 (tests/golden/h5/, generator oracle/gen_golden_h5.py; HydroRunBase.cpp:3308-3560 outputHdf5, 4818-5155 inputHdf5,
3823-4073 the XDMF index):
  * a run of this driver writes, at the same steps, files with the same datasets (every double equal), the same dataset
    layout (names, extents with / without ghosts) and attributes ("time step", "total time", nx, ny, nz, "ghost zone
    included") -- and the same .xmf index, byte for byte;
  * resumed from the REFERENCE's file of step N this driver ends in the state of the reference's file of the last step,
    every double equal (the reference's restart semantics: step count and time taken from the file);
  * restart_upscale from an .h5 of half the resolution.
libhdf5 is bound at run time (csrc/host/hdf5_io.cpp); these tests need it (the image has HDF5 1.10 under /opt/conda)."""
import ctypes as C
import json
import os
import shutil

import numpy as np
import pytest

import h5util
from conftest import ROOT, ini

H5DIR = os.path.join(ROOT, "tests", "golden", "h5")
CASES = json.load(open(os.path.join(H5DIR, "cases.json")))
pytestmark = pytest.mark.skipif(not h5util.available(), reason="no loadable libhdf5 on this machine")


def run(lib, base, ov, outdir, cwd):
    os.environ.pop("RGPU_RESTART_FORMAT", None)   # (tests/test_restart.py forces the raw dump through it)
    err = C.create_string_buffer(512)
    mc = C.c_double(0)
    old = os.getcwd()
    os.chdir(cwd)          # the .xmf index goes to the current directory, like the reference's
    try:
        n = lib.lib.rgpuh_run(ini(base).encode(), (ov + ";output.outputDir=%s" % outdir).encode(), C.byref(mc), err, 512)
    finally:
        os.chdir(old)
    assert n >= 0, err.value
    return n


def same_file(mine, ref, ghosts_meaningful=True):
    """ghosts_meaningful=False: a ghost-inclusive file of the PLAIN path.  Its ghost cells are whatever the step left there
    -- the reference's unguarded update loops scribble on them, this build copies the old values (values
    the reference computes but never consumes; the next step's ghost fill overwrites them) -- so only the interior is compared.
    On the rotating path the ghosts are filled at the END of the step and are part of the state: compared in full."""
    dm, am = h5util.read(mine)
    dr, ar = h5util.read(ref)
    assert sorted(dm) == sorted(dr)
    gw = 3 if len(dr) == 8 else 2
    for k in dr:
        assert dm[k].shape == dr[k].shape, (k, dm[k].shape, dr[k].shape)
        a, b = dm[k], dr[k]
        if not ghosts_meaningful and ar["ghost zone included"]:
            sl = (slice(gw, -gw),) * a.ndim
            a, b = a[sl], b[sl]
        assert np.array_equal(a, b), "%s: %d doubles differ" % (k, int((a != b).sum()))
    assert am == ar, (am, ar)


def ghosts_meaningful(name):
    return CASES[name]["base"] == "mhd_mri_3d"      # the only rotating-path case of the fixtures


def check_writer(lib, name, tmp_path):
    c = CASES[name]
    ov = c["overrides"] + ";run.nstepmax=%d;run.noutput=%d;run.tend=1e9;output.outputVtk=no;output.outputHdf5=yes;output.ghostIncluded=%s" % (
        c["last_step"], c["restart_step"], c["ghostIncluded"])
    assert run(lib, c["base"], ov, tmp_path, tmp_path) == c["last_step"]
    for s in (c["restart_step"], c["last_step"]):
        f = "%s_%07d.h5" % (c["prefix"], s)
        same_file(str(tmp_path / f), os.path.join(H5DIR, name, f), ghosts_meaningful(name))
    xmf = c["prefix"] + ".xmf"
    assert open(tmp_path / xmf).read() == open(os.path.join(H5DIR, name, xmf)).read()


def check_restart_from_reference_file(lib, name, tmp_path):
    c = CASES[name]
    f0 = "%s_%07d.h5" % (c["prefix"], c["restart_step"])
    shutil.copy(os.path.join(H5DIR, name, f0), tmp_path / f0)
    ov = c["overrides"] + ";run.nstepmax=%d;run.noutput=%d;run.tend=1e9;output.outputVtk=no;output.outputHdf5=yes;output.ghostIncluded=%s;run.restart=yes;run.restart_filename=%s" % (
        c["last_step"], c["restart_step"], c["ghostIncluded"], f0)
    assert run(lib, c["base"], ov, tmp_path, tmp_path) == c["last_step"]
    f1 = "%s_%07d.h5" % (c["prefix"], c["last_step"])
    same_file(str(tmp_path / f1), os.path.join(H5DIR, name, f1), ghosts_meaningful(name))


NAMES = sorted(CASES)


@pytest.mark.parametrize("name", NAMES)
def test_hdf5_files_equal_the_reference_emu(name, emu_lib, tmp_path):
    check_writer(emu_lib, name, tmp_path)


@pytest.mark.parametrize("name", NAMES)
def test_restart_from_the_reference_file_emu(name, emu_lib, tmp_path):
    check_restart_from_reference_file(emu_lib, name, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hdf5_files_equal_the_reference_gpu(name, gpu_lib, tmp_path):
    check_writer(gpu_lib, name, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_restart_from_the_reference_file_gpu(name, gpu_lib, tmp_path):
    check_restart_from_reference_file(gpu_lib, name, tmp_path)


def test_compression_level_and_rejections(emu_lib, tmp_path):
    """outputHdf5CompressionLevel changes the file, not the data; a file of another box / a text file are refused"""
    c = CASES["ot2d_16x12"]
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir(); b.mkdir()
    ov = c["overrides"] + ";run.nstepmax=2;run.noutput=2;output.outputVtk=no;output.outputHdf5=yes"
    run(emu_lib, c["base"], ov, a, a)
    run(emu_lib, c["base"], ov + ";output.outputHdf5CompressionLevel=6", b, b)
    f = "%s_0000002.h5" % c["prefix"]
    same_file(str(a / f), str(b / f))
    assert os.path.getsize(b / f) < os.path.getsize(a / f)
    err = C.create_string_buffer(512); mc = C.c_double(0)
    ov2 = "mesh.nx=24;mesh.ny=12;run.restart=yes;run.restart_filename=%s;output.outputDir=%s" % (f, a)
    assert emu_lib.lib.rgpuh_run(ini(c["base"]).encode(), ov2.encode(), C.byref(mc), err, 512) < 0 and b"another box" in err.value
    (a / "junk.h5").write_text("not an hdf5 file")
    ov3 = c["overrides"] + ";run.restart=yes;run.restart_filename=junk.h5;output.outputDir=%s" % a
    assert emu_lib.lib.rgpuh_run(ini(c["base"]).encode(), ov3.encode(), C.byref(mc), err, 512) < 0 and b"HDF5" in err.value
