"""The C-ABI library loads on a machine without a GPU and exports every symbol include/rgpu.h declares;
the device entry points refuse to run without a GPU (no CPU fallback in the product)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT, ini
from ramsesgpu_amd import _capi
from ramsesgpu_amd.solver import RgpuError, Solver, lib_path


def header_functions():
    text = open(os.path.join(ROOT, "include", "rgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rgpuh?_[a-z_0-9]+)\s*\(", text)))


def test_header_and_python_binding_agree():
    declared = header_functions()
    assert declared, "no functions parsed from include/rgpu.h"
    missing = [f for f in declared if f not in _capi.DECLARED_SYMBOLS]
    assert not missing, "functions declared in rgpu.h but unknown to the binding: %s" % missing


def test_library_exports_every_declared_symbol(product_lib):
    out = subprocess.check_output(["nm", "-D", "--defined-only", product_lib.path], universal_newlines=True)
    exported = set(l.split()[-1] for l in out.splitlines() if l.strip())
    for sym in header_functions():
        assert sym in exported, "%s is declared in include/rgpu.h but not exported by %s" % (sym, product_lib.path)


def test_contracted_variant_exports_the_same_abi(product_lib, contracted_lib):
    """librgpu_fast.so: same entry points, another arithmetic (rgpu_arithmetic() tells which library a process holds)"""
    out = subprocess.check_output(["nm", "-D", "--defined-only", contracted_lib.path], universal_newlines=True)
    exported = set(l.split()[-1] for l in out.splitlines() if l.strip())
    for sym in header_functions():
        assert sym in exported, "%s is declared in include/rgpu.h but not exported by %s" % (sym, contracted_lib.path)
    assert product_lib.arithmetic == "exact" and contracted_lib.arithmetic == "contracted"
    assert contracted_lib.backend == "hip-gfx950"
    comm = os.path.join(os.path.dirname(contracted_lib.path), "librgpu_comm_fast.so")
    needed = subprocess.check_output(["readelf", "-d", comm], universal_newlines=True)
    assert "librgpu_fast.so" in needed and "[librgpu.so]" not in needed      # the slab driver of the variant drives the variant


def test_params_struct_layout_matches_c(tmp_path):
    """sizeof / offsets of rgpu_params as seen by a C compiler == the ctypes mirror"""
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "rgpu.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n",'
                   'sizeof(rgpu_params),offsetof(rgpu_params,xMin),offsetof(rgpu_params,slope_type),'
                   'offsetof(rgpu_params,djet),offsetof(rgpu_params,nz_global));return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)], universal_newlines=True).split()]
    P = _capi.RgpuParams
    assert got == [C.sizeof(P), P.xMin.offset, P.slope_type.offset, P.djet.offset, P.nz_global.offset]


def test_product_is_the_hip_backend(product_lib):
    assert product_lib.backend == "hip-gfx950"
    assert os.path.samefile(product_lib.path, lib_path("exact"))


def test_no_cpu_fallback_without_gpu(product_lib):
    """On a machine without a GPU rgpu_create must fail with RGPU_ENODEVICE (-2), not compute on the host."""
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present: the failure path cannot be observed here")
    except ImportError:
        pass
    p = product_lib.params_from_ini(ini("orszag-tang"), "mesh.nx=16;mesh.ny=16")
    with pytest.raises(RgpuError) as e:
        Solver(p, product_lib)
    assert "(-2)" in str(e.value) and "no CPU fallback" in str(e.value)


def test_rejects_out_of_scope_configurations(emu_lib):
    L = emu_lib
    for base, ov, frag in (("orszag-tang", "MHD.implementationVersion=2", "implementationVersion"),
                           ("orszag-tang", "MHD.magRiemannSolver=roe", "magRiemannSolver"),
                           ("implode3d", "mesh.nz=16;hydro.slope_type=3", "slope_type"),      # hydro: slopes left unset by the reference
                           ("mhd_mri_3d", "mesh.nz=16;hydro.slope_type=3", "slope_type")):    # rotating 3D step: same
        p = L.params_from_ini(ini(base), "mesh.nx=16;mesh.ny=16;" + ov)
        with pytest.raises(RgpuError) as e:
            Solver(p, L)
        assert frag in str(e.value)
    for ov in ("gravity.self=yes", "hydro.scheme=plmde"):
        with pytest.raises(RgpuError):
            L.params_from_ini(ini("orszag-tang"), ov)


def test_create_with_null_params_returns_einval(product_lib, emu_lib):
    """rgpu_create(NULL, &ctx) reports RGPU_EINVAL through the returned context instead of dereferencing NULL"""
    for L in (product_lib, emu_lib):
        ctx = C.c_void_p()
        L.lib.rgpu_create.restype = C.c_int
        rc = L.lib.rgpu_create(None, C.byref(ctx))
        assert rc == -1 and ctx.value
        L.lib.rgpu_last_error.restype = C.c_char_p
        assert b"NULL" in L.lib.rgpu_last_error(ctx)
        L.lib.rgpu_destroy(ctx)
