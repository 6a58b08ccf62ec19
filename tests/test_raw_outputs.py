"""[output] outputXsm / outputNrrd / outputVtkAscii of the run driver against the files the reference binary wrote (tests/golden/raw/,
generator oracle/gen_golden_raw.py): same names, same bytes (Xsmurf: density as doubles behind a one-line header, in the
current directory; NRRD: every variable as 32-bit floats behind a text header, in the output directory; the .vti in text mode, 12 significant
digits)."""
import ctypes as C
import json
import os

import pytest

from conftest import ROOT, ini

RAW = os.path.join(ROOT, "tests", "golden", "raw")
CASES = json.load(open(os.path.join(RAW, "cases.json")))


def check(lib, name, tmp_path):
    c = CASES[name]
    ov = c["overrides"] + ";run.nstepmax=%d;run.noutput=%d;run.tend=1e9;output.outputVtk=yes;output.outputVtkAscii=yes;output.outputHdf5=no;output.outputXsm=yes;output.outputNrrd=yes;output.outputDir=%s" % (
        c["last_step"], c["last_step"], tmp_path)
    err = C.create_string_buffer(512); mc = C.c_double(0)
    old = os.getcwd()
    os.chdir(tmp_path)
    try:
        assert lib.lib.rgpuh_run(ini(c["base"]).encode(), ov.encode(), C.byref(mc), err, 512) == c["last_step"], err.value
    finally:
        os.chdir(old)
    ref = sorted(f for f in os.listdir(os.path.join(RAW, name)) if not f.endswith(".binary.vti"))
    assert len(ref) >= 5
    for f in ref:
        assert open(tmp_path / f, "rb").read() == open(os.path.join(RAW, name, f), "rb").read(), f
    # the default appended-raw .vti: the reference's bytes + one trailing comment line (step count and time for a restart)
    b = tmp_path / "bin"
    b.mkdir()
    ov2 = ov.replace("output.outputVtkAscii=yes", "output.outputVtkAscii=no").replace(str(tmp_path), str(b)).replace("outputXsm=yes", "outputXsm=no").replace("outputNrrd=yes", "outputNrrd=no")
    assert lib.lib.rgpuh_run(ini(c["base"]).encode(), ov2.encode(), C.byref(mc), err, 512) == c["last_step"], err.value
    for f in [f for f in os.listdir(os.path.join(RAW, name)) if f.endswith(".binary.vti")]:
        mine = open(b / f.replace(".binary.vti", ".vti"), "rb").read()
        want = open(os.path.join(RAW, name, f), "rb").read()
        assert mine[:len(want)] == want and mine[len(want):].startswith(b"<!-- rgpu restart: nStep=%d " % c["last_step"]) and mine.count(b"\n", len(want)) == 1


@pytest.mark.parametrize("name", sorted(CASES))
def test_xsm_and_nrrd_equal_the_reference_emu(name, emu_lib, tmp_path):
    check(emu_lib, name, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_xsm_and_nrrd_equal_the_reference_gpu(name, gpu_lib, tmp_path):
    check(gpu_lib, name, tmp_path)
