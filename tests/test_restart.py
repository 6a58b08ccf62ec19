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
    os.environ["RGPU_RESTART_FORMAT"] = "rgr"   # these tests cover the raw dump (the fallback without libhdf5); .h5: test_hdf5_io.py
    err = C.create_string_buffer(512)
    mc = C.c_double(0)
    full = ov + ";output.outputVtk=yes;output.outputHdf5=%s;output.ghostIncluded=yes;output.outputDir=%s" % (dump, outdir)
    try:
        n = lib.lib.rgpuh_run(ini(base).encode(), full.encode(), C.byref(mc), err, 512)
    finally:
        os.environ.pop("RGPU_RESTART_FORMAT", None)
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
    ha = open(a / files[2], "rb").read()[-200:].split(b"\n")[-2]     # the restart comment line after </VTKFile>
    hb = open(b / files[2], "rb").read()[-200:].split(b"\n")[-2]
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


# ---- [run] restart_upscale: resume on a mesh twice as fine (HydroRunBase::upscale, HydroRunBase.cpp:5170-5278) ----
def read_dump(path):
    """the raw restart dump of the run driver -> (header dict, array [nbVar, ksize, jsize, isize] incl. ghosts; zeros where
    the file holds the interior only)"""
    with open(path, "rb") as f:
        w = f.readline().split()
        assert w[0] == b"RGPU-RESTART" and w[1] == b"1"
        nx, ny, nz, gw, nv, gi, nstep = (int(x) for x in w[2:9])
        t = float.fromhex(w[9].decode())
        three_d = nz != 1
        shape = (nv, nz + 2 * gw if three_d else 1, ny + 2 * gw, nx + 2 * gw)
        if gi:
            a = np.frombuffer(f.read(), dtype="<f8").reshape(shape).copy()
        else:
            a = np.zeros(shape)
            inner = np.frombuffer(f.read(), dtype="<f8").reshape(nv, nz, ny, nx)
            ks = slice(gw, gw + nz) if three_d else slice(0, 1)
            a[:, ks, gw:gw + ny, gw:gw + nx] = inner
    return dict(nx=nx, ny=ny, nz=nz, gw=gw, nv=nv, ghosts=bool(gi), nstep=nstep, t=t), a


def read_state(path, three_d):
    """read_dump for both formats"""
    if path.endswith(".rgr"):
        return read_dump(path)
    import h5util
    d, at = h5util.read(path)
    names = [n for n in h5util.NAMES if n in d]
    gw = 3 if len(names) == 8 else 2
    gi = bool(at["ghost zone included"])
    nx, ny, nz = at["nx"], at["ny"], at["nz"]
    shape = (len(names), nz + 2 * gw if three_d else 1, ny + 2 * gw, nx + 2 * gw)
    a = np.zeros(shape)
    for v, n in enumerate(names):
        x = d[n] if three_d else d[n][None]
        if gi:
            a[v] = x
        else:
            ks = slice(gw, gw + nz) if three_d else slice(0, 1)
            a[v][ks, gw:gw + ny, gw:gw + nx] = x
    return dict(nx=nx, ny=ny, nz=nz, gw=gw, nv=len(names), ghosts=gi, nstep=at["time step"], t=at["total time"]), a


def upscale_expected(low, gw, three_d, mhd):
    """numpy statement of the rule: every fine cell (ghosts included) takes coarse cell (index + gw) // 2; the face field on
    a fine face in the middle of a coarse cell is the mean of the two coarse faces along its own direction"""
    nv, lk, lj, li = low.shape
    isz, jsz = 2 * (li - 2 * gw) + 2 * gw, 2 * (lj - 2 * gw) + 2 * gw
    ksz = 2 * (lk - 2 * gw) + 2 * gw if three_d else 1
    I = (np.arange(isz) + gw) // 2; J = (np.arange(jsz) + gw) // 2
    K = (np.arange(ksz) + gw) // 2 if three_d else np.zeros(1, dtype=int)
    out = low[:, K][:, :, J][:, :, :, I].copy()
    if mhd:
        oi = (np.arange(isz) + gw) % 2 == 1
        out[5][:, :, oi] = (low[5][K][:, J][:, :, I[oi]] + low[5][K][:, J][:, :, I[oi] + 1]) / 2
        oj = (np.arange(jsz) + gw) % 2 == 1
        out[6][:, oj, :] = (low[6][K][:, J[oj]][:, :, I] + low[6][K][:, J[oj] + 1][:, :, I]) / 2
        if three_d:
            ok = (np.arange(ksz) + gw) % 2 == 1
            out[7][ok] = (low[7][K[ok]][:, J][:, :, I] + low[7][K[ok] + 1][:, J][:, :, I]) / 2
    return out


UPSCALE_CASES = [
    ("orszag-tang", "mesh.nx=%d;mesh.ny=%d", (12, 8, 1), "yes"),
    ("orszag-tang", "mesh.nx=%d;mesh.ny=%d", (12, 8, 1), "no"),
    ("orszag-tang3d", "mesh.nx=%d;mesh.ny=%d;mesh.nz=%d", (6, 8, 6), "yes"),
    ("implode3d", "mesh.nx=%d;mesh.ny=%d;mesh.nz=%d;hydro.riemannSolver=hllc", (6, 6, 8), "yes"),
    ("jet2d_cpu", "mesh.nx=%d;mesh.ny=%d;jet.ijet=2;jet.offsetJet=1", (8, 20, 1), "no"),
]


def check_upscale(lib, base, ovf, dims, ghosts, tmp_path, fmt="rgr"):
    """fmt: "rgr" = the raw dump, "h5" = the reference's HDF5 format (needs libhdf5)"""
    old = os.getcwd()
    os.chdir(tmp_path)      # (the .xmf index of HDF5 outputs goes to the current directory)
    try:
        _check_upscale(lib, base, ovf, dims, ghosts, tmp_path, fmt)
    finally:
        os.chdir(old)
        os.environ.pop("RGPU_RESTART_FORMAT", None)


def _check_upscale(lib, base, ovf, dims, ghosts, tmp_path, fmt):
    a, b = tmp_path / "coarse", tmp_path / "fine"
    a.mkdir(); b.mkdir()
    d = dims[:2] if dims[2] == 1 else dims
    common = ";run.noutput=1;run.tend=1e9;output.outputVtk=no;output.outputHdf5=yes;output.ghostIncluded=%s" % ghosts
    err = C.create_string_buffer(512); mc = C.c_double(0)
    os.environ["RGPU_RESTART_FORMAT"] = fmt
    ext = ".h5" if fmt == "h5" else ".rgr"
    ov = (ovf % d) + common + ";run.nstepmax=4;output.outputDir=%s" % a
    assert lib.lib.rgpuh_run(ini(base).encode(), ov.encode(), C.byref(mc), err, 512) == 4, err.value
    src = sorted(f for f in os.listdir(a) if f.endswith("0000004" + ext))[0]
    (b / src).write_bytes((a / src).read_bytes())
    hl, low = read_state(str(a / src), dims[2] != 1)
    assert hl["nstep"] == 4 and hl["ghosts"] == (ghosts == "yes")
    fine = tuple(2 * x for x in d)
    ov = (ovf % fine) + common + ";run.nstepmax=3;run.restart=yes;run.restart_upscale=yes;run.restart_filename=%s;output.outputDir=%s" % (src, b)
    assert lib.lib.rgpuh_run(ini(base).encode(), ov.encode(), C.byref(mc), err, 512) == 3, err.value
    h0, u0 = read_state(str(b / [f for f in os.listdir(b) if f.endswith("0000000" + ext)][0]), dims[2] != 1)
    assert h0["nstep"] == 0 and h0["t"] == hl["t"]          # the time of the coarse run, a fresh step count
    gw, three_d, mhd = hl["gw"], dims[2] != 1, hl["nv"] == 8
    want = upscale_expected(low, gw, three_d, mhd)
    assert want.shape == u0.shape
    if ghosts == "yes":      # every cell comes from the file: no boundary fill at the start (MHDRunGodunov.cpp:3818-3824)
        assert np.array_equal(u0, want)
    else:
        ks = slice(gw, -gw) if three_d else slice(0, 1)
        assert np.array_equal(u0[:, ks, gw:-gw, gw:-gw], want[:, ks, gw:-gw, gw:-gw])
    if mhd and ghosts == "yes":   # the interpolation keeps the discrete divergence of the face field at round-off (the high
                                  # faces of the last cells are ghost-side values: in the file only with ghostIncluded)
        bx, by, bz = u0[5], u0[6], u0[7]
        ks = slice(gw, -gw) if three_d else slice(0, 1)
        div = (bx[ks, gw:-gw, gw + 1:-gw + 1] - bx[ks, gw:-gw, gw:-gw]) * fine[0] + (by[ks, gw + 1:-gw + 1, gw:-gw] - by[ks, gw:-gw, gw:-gw]) * fine[1]
        if three_d:
            div = div + (bz[gw + 1:-gw + 1, gw:-gw, gw:-gw] - bz[gw:-gw, gw:-gw, gw:-gw]) * fine[2]
        assert np.abs(div).max() < 1e-10 * max(1.0, np.abs(bx).max() * max(fine))
    h3, u3 = read_state(str(b / [f for f in os.listdir(b) if f.endswith("0000003" + ext)][0]), dims[2] != 1)
    assert h3["nstep"] == 3 and h3["t"] > hl["t"] and np.isfinite(u3).all()


UIDS = ["%s-ghosts_%s" % (c[0], c[3]) for c in UPSCALE_CASES]


@pytest.mark.parametrize("base,ovf,dims,ghosts", UPSCALE_CASES, ids=UIDS)
def test_restart_upscale_emu(base, ovf, dims, ghosts, emu_lib, tmp_path):
    check_upscale(emu_lib, base, ovf, dims, ghosts, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("base,ovf,dims,ghosts", UPSCALE_CASES[:3], ids=UIDS[:3])
def test_restart_upscale_gpu(base, ovf, dims, ghosts, gpu_lib, tmp_path):
    check_upscale(gpu_lib, base, ovf, dims, ghosts, tmp_path)


@pytest.mark.parametrize("base,ovf,dims,ghosts", UPSCALE_CASES[:4], ids=UIDS[:4])
def test_restart_upscale_from_hdf5_emu(base, ovf, dims, ghosts, emu_lib, tmp_path):
    import h5util
    if not h5util.available():
        pytest.skip("no loadable libhdf5 on this machine")
    check_upscale(emu_lib, base, ovf, dims, ghosts, tmp_path, fmt="h5")
