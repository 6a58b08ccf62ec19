"""The reference-side binding of INTEGRATION.md (integration/MHDRunGodunovHip.h, HydroRunGodunovHip.h).

not gpu:  the stub headers compile against the UNMODIFIED reference headers when /root/reference is present (skipped on
          the GPU box, where it is not); the binary built from them, oracle/_ref/euler_ref_hip = the reference's own run
          classes / start() loop / VTK writer linked with librgpu.so, refuses to run without a GPU (no CPU fallback).
gpu:      that binary reproduces the golden fixtures of euler_cpu bit for bit -- the drop-in claim, end to end."""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from conftest import ROOT, golden_cases, load_golden

REF_SRC = "/root/reference/src"
BIN = os.path.join(ROOT, "oracle", "_ref", "euler_ref_hip")
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _includes():
    inc = [REF_SRC, REF_SRC + "/utils", REF_SRC + "/utils/config/inih", REF_SRC + "/hydro", REF_SRC + "/utils/config",
           REF_SRC + "/utils/monitoring", REF_SRC + "/utils/cnpy", os.path.join(ROOT, "include"), os.path.join(ROOT, "integration")]
    return [a for i in inc for a in ("-I", i)]


@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="reference sources are not on this machine")
def test_stub_headers_compile_against_reference_headers(tmp_path):
    """signature drift guard: member names, virtual signatures (make_all_boundaries(HostArray<real_t>&), copyGpuToCpu(int),
    oneStepIntegration(int&, real_t&, real_t&)), abstractness -- g++ -fsyntax-only with the reference's own flags"""
    src = tmp_path / "check.cpp"
    src.write_text('#include "MHDRunGodunovHip.h"\n#include "HydroRunGodunovHip.h"\n'
                   "hydroSimu::HydroRunBase* make_mhd(ConfigMap& c) { return new hydroSimu::MHDRunGodunovHip(c); }\n"
                   "hydroSimu::HydroRunBase* make_hydro(ConfigMap& c) { return new hydroSimu::HydroRunGodunovHip(c); }\n")
    res = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-DUSE_DOUBLE", "-w"] + _includes() + [str(src)],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    assert res.returncode == 0, res.stdout[-3000:]


def _run_case(name, tmp):
    from gen_golden import apply_overrides, read_vti, VAR_NAMES
    case = golden_cases()[name]
    ini = open(os.path.join(ROOT, "configs", case["base"] + ".ini")).read()
    ini = apply_overrides(ini, case["overrides"] + ";run.nlog=1;output.outputVtk=yes;output.outputHdf5=no;output.outputDir=./")
    prefix = re.search(r"outputPrefix=(\S+)", ini).group(1)
    open(os.path.join(tmp, "case.ini"), "w").write(ini)
    res = subprocess.run([BIN, "--param", "case.ini"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         universal_newlines=True, timeout=600)
    out = {}
    if res.returncode == 0:
        for s in case["steps"]:
            fields, _ = read_vti(os.path.join(tmp, "%s_%07d.vti" % (prefix, s)))
            out[s] = np.stack([fields[n] for n in VAR_NAMES[len(fields)]])
    return res, out, case


@pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/euler_ref_hip was not built (needs the reference sources at build time)")
def test_reference_driver_on_librgpu_has_no_cpu_fallback(tmp_path):
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present: the failure path cannot be observed here")
    except ImportError:
        pass
    res, _, _ = _run_case("implode3d_16_hllc", str(tmp_path))
    assert res.returncode == 1 and "no CPU fallback" in res.stdout, res.stdout[-2000:]


GOLDEN_VIA_REFERENCE_DRIVER = ["ot2d_64", "mri_16x32x16", "implode3d_16_hllc", "jet2d_20x80", "ot3d_16", "briowu_x_64", "ot3d_12_visc_res",
                               "mri_strat_8x16x32", "kepler3d_16x16x6"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n in GOLDEN_VIA_REFERENCE_DRIVER if n in golden_cases()])
def test_reference_driver_on_librgpu_reproduces_euler_cpu(name, tmp_path):
    """ConfigMap, init_simulation, start(), outputVtk are the reference's code; oneStepIntegration / ghost fill / host mirror
    go through integration/*.h to librgpu.so: the .vti files equal the ones euler_cpu wrote (tests/golden)"""
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/euler_ref_hip was not built")
    res, out, case = _run_case(name, str(tmp_path))
    assert res.returncode == 0, res.stdout[-3000:]
    assert "backend : hip-gfx950" in res.stdout
    g = load_golden(name)
    for s in case["steps"]:
        ref = g["step_%d" % s]
        assert out[s].shape == ref.shape
        nbad = int((out[s] != ref).sum())
        assert nbad == 0, "%s step %d: %d of %d doubles differ from euler_cpu" % (name, s, nbad, ref.size)
