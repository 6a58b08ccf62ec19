"""Shared fixtures.  Markers: `gpu` = needs a real MI355X (run on the GPU box), everything else runs on CPU.

Layers under test
  * oracle        oracle/liboracle.so        CPU restatement of the reference (the checker)
  * host          ramsesgpu_amd/librgpu.so   product library: host entry points work without a GPU,
                                             device entry points must fail with RGPU_ENODEVICE there
  * emu           tests/_build/librgpu_emu.so  TEST-ONLY build of the same sources with tests/emu/rg_backend.h
                                             (kernel bodies executed by host loops) -- never shipped
  * gpu           ramsesgpu_amd/librgpu.so   on a GPU box: the HIP path, the thing being verified
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")
CONFIGS = os.path.join(ROOT, "configs")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X GPU (run with -m gpu on the GPU box)")


def ini(base):
    return os.path.join(CONFIGS, base + ".ini")


def golden_cases():
    return json.load(open(os.path.join(GOLDEN, "cases.json")))


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def _run(cmd, **kw):
    subprocess.check_call(cmd, **kw)


@pytest.fixture(scope="session")
def oracle():
    from oracle_api import Oracle
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    srcs = [os.path.join(ROOT, "oracle", "restate", f) for f in os.listdir(os.path.join(ROOT, "oracle", "restate"))]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        _run(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    return Oracle(so)


@pytest.fixture(scope="session")
def product_lib():
    """the shipped library (HIP backend).  Built in-tree by __graft_entry__.build()."""
    from ramsesgpu_amd.solver import Library, lib_path
    if not os.path.exists(lib_path("exact")):
        import __graft_entry__
        __graft_entry__.build()
    return Library(lib_path("exact"))


@pytest.fixture(scope="session")
def contracted_lib():
    """the opt-in variant of the product with contracted arithmetic (librgpu_fast.so, see rgpu_arithmetic() in rgpu.h)"""
    from ramsesgpu_amd.solver import Library, lib_path
    if not os.path.exists(lib_path("contracted")):
        import __graft_entry__
        __graft_entry__.build()
    return Library(lib_path("contracted"))


@pytest.fixture(scope="session")
def emu_lib():
    """TEST-ONLY host emulation build of the device sources (see tests/emu/rg_backend.h)."""
    from ramsesgpu_amd.solver import Library
    out_dir = os.path.join(ROOT, "tests", "_build")
    so = os.path.join(out_dir, "librgpu_emu.so")
    csrc = os.path.join(ROOT, "ramsesgpu_amd", "csrc")
    deps = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".h", ".cpp"))]
    deps += [os.path.join(csrc, "host", f) for f in os.listdir(os.path.join(csrc, "host"))]
    deps += [os.path.join(csrc, "api", f) for f in os.listdir(os.path.join(csrc, "api"))]
    deps += [os.path.join(ROOT, "tests", "emu", "rg_backend.h"), os.path.join(ROOT, "include", "rgpu.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        os.makedirs(out_dir, exist_ok=True)
        host = [os.path.join(csrc, "host", f) for f in ("ini_config.cpp", "host_params.cpp", "init_conditions.cpp", "host_capi.cpp", "run_driver.cpp", "hdf5_io.cpp")]
        _run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wl,-Bsymbolic-functions", "-Wno-unknown-pragmas", "-I", os.path.join(ROOT, "tests", "emu"), "-I", csrc,
              os.path.join(csrc, "rgpu_api.cpp")] + host + ["-ldl", "-o", so])
    lib = Library(so)
    assert "emulation" in lib.backend
    return lib


def _apply_test_options(lib):
    """RGPU_TEST_OPTIONS="ghost_images=0,spec=0": the whole GPU suite with diagnostic options of the library switched (a variable of the
    TESTS, not of the product: the library itself has rgpu_set_option, include/rgpu.h)"""
    for kv in filter(None, os.environ.get("RGPU_TEST_OPTIONS", "").split(",")):
        k, v = kv.split("=")
        lib.set_option(k.strip(), int(v))


@pytest.fixture(scope="session")
def gpu_lib(product_lib):
    """the product library on a machine that really has a GPU; fails loudly otherwise"""
    import ctypes as C
    from ramsesgpu_amd.solver import Solver
    p = product_lib.params_from_ini(ini("orszag-tang"), "mesh.nx=8;mesh.ny=8")
    Solver(p, product_lib).close()  # raises RgpuError(RGPU_ENODEVICE) without a GPU: no silent fallback
    _apply_test_options(product_lib)
    return product_lib


@pytest.fixture(scope="session")
def gpu_contracted_lib(contracted_lib):
    from ramsesgpu_amd.solver import Solver
    p = contracted_lib.params_from_ini(ini("orszag-tang"), "mesh.nx=8;mesh.ny=8")
    Solver(p, contracted_lib).close()
    _apply_test_options(contracted_lib)
    return contracted_lib
