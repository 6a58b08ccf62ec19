"""The C++ z-slab driver (include/rgpu_comm.h, csrc/comm/rgpu_comm.cpp).

not gpu:  librgpu_comm.so loads and exports every symbol the header declares; world sizes 1, 2, 3 on CPU through the driver's
          real schedule with the TEST-ONLY callback transport (gloo underneath): == single-domain oracle, bit for bit.
gpu:      nranks = 1 on the GPU box through the product libraries: the RCCL code path (ncclCommInitRank, grouped
          ncclSend / ncclRecv to itself on the halo stream, event ordering, ncclAllReduce-free dt) with periodic z."""
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, ini
from ramsesgpu_amd import comm as rcomm


def header_functions():
    text = open(os.path.join(ROOT, "include", "rgpu_comm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rgpuh?_[a-z_0-9]+)\s*\(", text)))


def test_comm_library_exports_every_declared_symbol(product_lib):
    declared = header_functions()
    assert declared and set(declared) == set(rcomm.DECLARED_SYMBOLS), (declared, rcomm.DECLARED_SYMBOLS)
    out = subprocess.check_output(["nm", "-D", "--defined-only", rcomm.comm_lib_path()], universal_newlines=True)
    exported = set(l.split()[-1] for l in out.splitlines() if l.strip())
    for sym in declared:
        assert sym in exported, "%s is declared in include/rgpu_comm.h but not exported" % sym
    CL = rcomm.load_comm_library()     # loads on a machine without GPUs too (no compute call is made)
    assert CL.rgpu_comm_transport_name() == b"rccl"


@pytest.fixture(scope="session")
def comm_emu_lib(emu_lib):
    """TEST-ONLY build of the driver against the callback transport and the emulation library"""
    out_dir = os.path.join(ROOT, "tests", "_build")
    so = os.path.join(out_dir, "librgpu_comm_emu.so")
    src = os.path.join(ROOT, "ramsesgpu_amd", "csrc", "comm", "rgpu_comm.cpp")
    deps = [src, os.path.join(ROOT, "tests", "emu", "rg_transport.h"), os.path.join(ROOT, "include", "rgpu_comm.h"),
            os.path.join(out_dir, "librgpu_emu.so")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wl,-Bsymbolic-functions", "-I", os.path.join(ROOT, "tests", "emu"), src,
                               "-L", out_dir, "-lrgpu_emu", "-Wl,-rpath,$ORIGIN", "-o", so])
    return so


OPEN_BC = ";mesh.boundary_xmin=2;mesh.boundary_xmax=2;mesh.boundary_ymin=1;mesh.boundary_ymax=2;mesh.boundary_zmin=2;mesh.boundary_zmax=1"
# (ini, overrides, steps, world, overlap)
CASES = [
    ("mhd_mri_3d", "mesh.nx=8;mesh.ny=12;mesh.nz=12;MHD.omega0=0.02", 3, 1, 1),        # self ring: the rank is its own neighbour
    ("orszag-tang3d", "mesh.nx=8;mesh.ny=8;mesh.nz=16", 3, 1, 1),
    ("implode3d", "mesh.nx=8;mesh.ny=8;mesh.nz=12;hydro.riemannSolver=hllc", 3, 1, 1),  # no slab interface at all
    ("orszag-tang3d", "mesh.nx=8;mesh.ny=8;mesh.nz=12", 3, 2, 1),                      # plain MHD, periodic ring, no inner planes
    ("orszag-tang3d", "mesh.nx=8;mesh.ny=8;mesh.nz=24", 4, 2, 1),                      # ... with inner planes
    ("orszag-tang3d", "mesh.nx=8;mesh.ny=8;mesh.nz=24", 3, 2, 0),                      # serial schedule
    ("orszag-tang3d", "mesh.nx=8;mesh.ny=8;mesh.nz=20" + OPEN_BC, 4, 2, 1),            # open / reflecting faces: dt sees unfilled ghosts
    ("mhd_mri_3d", "mesh.nx=8;mesh.ny=12;mesh.nz=20;MHD.omega0=0.02", 4, 2, 1),        # rotating + shearing box
    ("mhd_mri_3d", "mesh.nx=8;mesh.ny=12;mesh.nz=12;MHD.omega0=0.02", 3, 2, 0),
    ("implode3d", "mesh.nx=8;mesh.ny=8;mesh.nz=16", 4, 2, 1),                          # hydro, Dirichlet ends
    ("mhd_mri_3d", "mesh.nx=6;mesh.ny=8;mesh.nz=27", 3, 3, 1),                         # three slabs, inner planes
    ("mhd_mri_3d", "mesh.nx=6;mesh.ny=8;mesh.nz=12", 3, 3, 1),                         # three slabs thinner than 2 gw
    ("orszag-tang3d", "mesh.nx=6;mesh.ny=6;mesh.nz=18;hydro.nu=0.005;MHD.eta=0.01", 3, 3, 1),   # dissipative stage: second exchange
    ("turbulence_hydro", "mesh.nx=8;mesh.ny=8;mesh.nz=12", 3, 2, 1),                   # random forcing: SUM all-reduce
    ("mhd_mri_3d_stratified", "mesh.nx=6;mesh.ny=8;mesh.nz=24;hydro.slope_type=2.0;MRI.amp=0.3", 3, 2, 1),
    # three slabs of the stratified box: the end slabs cannot carry the CFL scan in their update kernels (stratified z face on
    # the rotating path), the inner one could -- the ranks must agree (rgpu_inv_dt_fusable) or the 1/dt all-reduce mismatches
    ("mhd_mri_3d_stratified", "mesh.nx=6;mesh.ny=8;mesh.nz=27;hydro.slope_type=2.0;MRI.amp=0.3", 3, 3, 1),
    ("turbulence_hydro_ou", "mesh.nx=8;mesh.ny=8;mesh.nz=12;turbulence-Ornstein-Uhlenbeck.initialDensityPerturbationAmplitude=0.1", 3, 2, 1),   # Ornstein-Uhlenbeck forcing: same process on every rank
    ("turbulence_mhd_ou", "mesh.nx=6;mesh.ny=6;mesh.nz=18", 3, 3, 1),
    # schedule 2, boundary-first: fluxes + update of the boundary planes, exchange, THEN the sweep of the inner planes (3D MHD)
    ("mhd_mri_3d", "mesh.nx=8;mesh.ny=12;mesh.nz=40;MHD.omega0=0.02", 4, 2, 2),       # rotating + shearing box: shear remap per flux range
    ("orszag-tang3d", "mesh.nx=8;mesh.ny=8;mesh.nz=40", 4, 2, 2),
    ("orszag-tang3d", "mesh.nx=8;mesh.ny=8;mesh.nz=32" + OPEN_BC, 4, 2, 2),           # end slabs with physical z faces
    ("mhd_mri_3d", "mesh.nx=6;mesh.ny=8;mesh.nz=60", 3, 3, 2),
    ("mhd_mri_3d", "mesh.nx=8;mesh.ny=12;mesh.nz=24;MHD.omega0=0.02", 3, 1, 2),       # ring of one
    ("implode3d", "mesh.nx=8;mesh.ny=8;mesh.nz=40", 3, 2, 2),                          # hydro: the sweep is the whole step, same as 1
]


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_worker(base, ov, nsteps, world, overlap, tmp_path, env_extra=None, timeout=300):
    out = str(tmp_path / "result.txt")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "comm_worker.py"), base, ov, str(nsteps), out]
    env = dict(os.environ, OMP_NUM_THREADS="1", COMM_OVERLAP=str(overlap), **(env_extra or {}))
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=timeout)
    assert res.returncode == 0, res.stdout[-3000:]
    assert open(out).read().split()[0] == "OK", open(out).read()
    return open(out).read().split()[1:]


# The GPU suite's small cases: interpreter start, the torch import and the HIP context of every rank process are most of a case's wall
# time, so the cases of one test function that share a rank count run in ONE set of rank processes (comm_worker.py --batch), launched
# by the first test of the group; every test then reads its own case's result file.
_BATCHES = {}


def run_batched(group, cases, case, tmp_path_factory, env_extra=None, nsteps=None, timeout=900):
    """cases: (base, ov, nsteps, world, overlap) tuples of one test function; `case` is the one asked about.  nsteps overrides the tuple's."""
    import json
    world = case[3]
    key = (group, world)
    if key not in _BATCHES:
        d = tmp_path_factory.mktemp("batch_%s_x%d" % (group, world))
        mine = [c for c in cases if c[3] == world]
        spec = [{"argv": [c[0], c[1], str(nsteps or c[2]), str(d / ("result_%d.txt" % n))], "env": {"COMM_OVERLAP": str(c[4])}} for n, c in enumerate(mine)]
        (d / "spec.json").write_text(json.dumps(spec))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
               os.path.join(ROOT, "tests", "comm_worker.py"), "--batch", str(d / "spec.json")]
        env = dict(os.environ, OMP_NUM_THREADS="1", **(env_extra or {}))
        try:
            res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=timeout)
            log = res.stdout[-3000:]
        except subprocess.TimeoutExpired as e:
            log = "TIMEOUT after %d s\n%s" % (timeout, (e.stdout or "")[-3000:])
        _BATCHES[key] = ({c: (d / ("result_%d.txt" % n)) for n, c in enumerate(mine)}, log)
    files, log = _BATCHES[key]
    f = files[case]
    assert f.exists(), "no result for this case; the batch's output:\n" + log
    text = f.read_text()
    assert text.split()[0] == "OK", text + "\n" + log
    return text.split()[1:]


@pytest.mark.parametrize("base,ov,nsteps,world,overlap", CASES,
                         ids=["%s-%d-x%d-%s" % (c[0], n, c[3], ("serial", "overlap", "boundary-first")[c[4]]) for n, c in enumerate(CASES)])
def test_cpp_driver_matches_single_domain(base, ov, nsteps, world, overlap, comm_emu_lib, oracle, tmp_path):
    run_worker(base, ov, nsteps, world, overlap, tmp_path)


@pytest.mark.parametrize("base,ov,nsteps,world", [("mhd_mri_3d", "mesh.nx=6;mesh.ny=8;mesh.nz=24;MHD.omega0=0.02", 3, 2),     # thin MHD slabs: boundary-first
                                                  ("implode3d", "mesh.nx=8;mesh.ny=8;mesh.nz=16;hydro.riemannSolver=hllc", 3, 2)],   # hydro: overlapped
                         ids=["mhd-thin-slabs", "hydro"])
def test_default_schedule_follows_the_slab_thickness(base, ov, nsteps, world, comm_emu_lib, oracle, tmp_path):
    """rgpu_comm_set_overlap(-1): the driver picks the schedule (comm_worker.py asserts rgpu_comm_schedule) -- and the run equals the oracle"""
    run_worker(base, ov, nsteps, world, -1, tmp_path)


RUN_STEPS = [CASES[n] for n in (0, 2, 4, 6, 7, 9, 10, 11, 12, 14, 18, 19, 20, 21, 22, 23)] + [
    ("mhd_mri_3d", "mesh.nx=6;mesh.ny=8;mesh.nz=32;MHD.omega0=0.02", 3, 4, 1),      # four slabs
    ("mhd_mri_3d", "mesh.nx=6;mesh.ny=8;mesh.nz=64;MHD.omega0=0.02", 3, 4, 2)]


@pytest.mark.parametrize("base,ov,nsteps,world,overlap", RUN_STEPS,
                         ids=["%s-%d-x%d-%s" % (c[0], n, c[3], ("serial", "overlap", "boundary-first")[c[4]]) for n, c in enumerate(RUN_STEPS)])
def test_cpp_driver_run_steps_matches_single_domain(base, ov, nsteps, world, overlap, comm_emu_lib, oracle, tmp_path):
    """the same runs through rgpu_comm_run_steps (one plain step, then the rest as one batch: per step the 1/dt slots all-reduced in
    place, the clock record formed from them, the step pieces reading it -- csrc/step_clock_rec.h; the emulation backend forms the same
    records) and with an end time inside the batch: state, dt sequence, step count and t equal the single-domain oracle's"""
    run_worker(base, ov, nsteps + 1, world, overlap, tmp_path, env_extra={"COMM_RUN_STEPS": "1"})


@pytest.mark.parametrize("base,ov,world,fail_rank,fail_step,overlap", [
    ("orszag-tang3d", "mesh.nx=8;mesh.ny=8;mesh.nz=24", 2, 1, 0, 1),     # step 0: the failing rank has only ever all-reduced after a FULL scan
    ("orszag-tang3d", "mesh.nx=8;mesh.ny=8;mesh.nz=24", 2, 0, 2, 1),     # steady state of the overlapped schedule (fused scan)
    ("mhd_mri_3d", "mesh.nx=6;mesh.ny=8;mesh.nz=27;MHD.omega0=0.02", 3, 1, 1, 1),   # three slabs, rotating path
    ("orszag-tang3d", "mesh.nx=8;mesh.ny=8;mesh.nz=24", 2, 1, 1, 0),     # serial schedule
], ids=["step0", "steady", "x3-rotating", "serial"])
@pytest.mark.parametrize("loop", ["host", "batch", "single"])
def test_a_failed_step_piece_on_one_rank_reaches_every_rank(base, ov, world, fail_rank, fail_step, overlap, loop, comm_emu_lib, tmp_path):
    """a launch error inside godunov_unsplit on ONE rank: that rank still posts the exchange, poisons the next 1/dt all-reduce with
    +inf -- always RGPU_DT_SLOTS values, so its size cannot differ from the healthy ranks' -- and every rank returns an error"""
    out = str(tmp_path / "result.txt")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "comm_worker.py"), "--poison", base, ov, str(fail_rank), str(fail_step), out]
    env = dict(os.environ, OMP_NUM_THREADS="1", COMM_OVERLAP=str(overlap))
    if loop != "host":   # the library's own loop (rgpu_comm_run_steps): failure in the first step of a call of three / in a call of one
        env["POISON_RUN_STEPS"] = loop
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=120)
    assert res.returncode == 0, res.stdout[-3000:]
    assert open(out).read().strip() == "OK", open(out).read()


# ---- the z-slab FRONT END (euler_hip --slabs N = rgpuh_run_slabs): run loop, one HDF5 file of the whole box per output step
# written slab after slab, restart of every slab from such a file -- against the single-domain front end ----
FRONTEND_CASES = [
    ("mhd_mri_3d", "mesh.nx=6;mesh.ny=8;mesh.nz=12;MRI.amp=0.2;run.nstepmax=6;run.noutput=3;run.tend=1e9;output.outputVtk=no;output.outputHdf5=yes;output.ghostIncluded=yes", 2),
    ("mhd_mri_3d", "mesh.nx=6;mesh.ny=8;mesh.nz=18;MRI.amp=0.2;run.nstepmax=4;run.noutput=2;run.tend=1e9;output.outputVtk=no;output.outputHdf5=yes;output.ghostIncluded=no", 3),
    ("implode3d", "mesh.nx=8;mesh.ny=8;mesh.nz=12;hydro.riemannSolver=hllc;run.nstepmax=4;run.noutput=2;run.tend=1e9;output.outputVtk=no;output.outputHdf5=yes;output.ghostIncluded=no", 2),
    ("orszag-tang3d", "mesh.nx=8;mesh.ny=8;mesh.nz=16;run.nstepmax=4;run.noutput=2;run.tend=1e9;output.outputVtk=no;output.outputHdf5=yes;output.outputHdf5CompressionLevel=4", 2),
    # + the history file of the MRI run: global sums through the slab driver (rgpu_comm_history_mri), written by rank 0
    ("mhd_mri_3d", "mesh.nx=6;mesh.ny=8;mesh.nz=12;MRI.amp=0.2;run.nstepmax=6;run.noutput=3;run.tend=1e9;output.outputVtk=no;output.outputHdf5=yes;history.enabled=yes;history.dtHist=1e-9", 3),
]


def run_frontend(base, ov, world, outdir, tmp_path, timeout=300, env_extra=None):
    out = str(tmp_path / "result.txt")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "comm_worker.py"), "--frontend", base, ov, str(outdir), out]
    env = dict(os.environ, OMP_NUM_THREADS="1", **(env_extra or {}))
    env.pop("RGPU_RESTART_FORMAT", None)
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=timeout)
    assert res.returncode == 0, res.stdout[-3000:]
    assert open(out).read().strip() == "OK", open(out).read()


@pytest.mark.parametrize("base,ov,world", FRONTEND_CASES, ids=["%s-x%d-%d" % (c[0], c[2], n) for n, c in enumerate(FRONTEND_CASES)])
def test_slab_front_end_writes_the_single_domain_files(base, ov, world, comm_emu_lib, tmp_path):
    import h5util
    if not h5util.available():
        pytest.skip("no loadable libhdf5 on this machine")
    run_frontend(base, ov, world, tmp_path / "run", tmp_path)


@pytest.mark.parametrize("broken_rank", [0, 2])
def test_slab_front_end_fails_on_every_rank_when_one_rank_cannot_write(broken_rank, comm_emu_lib, tmp_path):
    """one rank alone cannot load libhdf5: the ranks agree on the outcome of the output step (hooks.agree) and all return
    an error -- nobody is left waiting in a collective (the subprocess timeout would catch a hang)"""
    import h5util
    if not h5util.available():
        pytest.skip("no loadable libhdf5 on this machine")
    base, ov, world = FRONTEND_CASES[1]
    run_frontend(base, ov, world, tmp_path / "run", tmp_path, timeout=120, env_extra={"COMM_BREAK_HDF5_ON_RANK": str(broken_rank)})


def test_slab_front_end_writes_pvti_and_pieces(comm_emu_lib, tmp_path):
    """[output] outputVtk=yes in a z-slab run: per-rank .vti (own planes + the overlap plane) and the .pvti index of
    HydroRunBaseMpi::outputVtk; the pieces hold the doubles of the single-domain .vti files"""
    ov = "mesh.nx=6;mesh.ny=8;mesh.nz=18;MRI.amp=0.2;run.nstepmax=4;run.noutput=2;run.tend=1e9;output.outputVtk=yes;output.outputHdf5=no"
    run_frontend("mhd_mri_3d", ov, 3, tmp_path / "run", tmp_path)


def test_slab_front_end_writes_the_mpi_turbulence_history(comm_emu_lib, tmp_path):
    """history file of a z-slab run of the MHD turbulence problem: the 16-column row of the MPI classes (global sums through
    rgpu_comm_history_turbulence), checked against a numpy statement of that row on the single-domain states"""
    import h5util
    if not h5util.available():
        pytest.skip("no loadable libhdf5 on this machine")
    ov = ("mesh.nx=6;mesh.ny=6;mesh.nz=12;run.nstepmax=3;run.noutput=1;run.tend=1e9;output.outputVtk=no;output.outputHdf5=yes;"
          "output.ghostIncluded=yes;history.enabled=yes;history.dtHist=1e-12")
    run_frontend("turbulence_mhd_ou", ov, 2, tmp_path / "run", tmp_path, env_extra={"COMM_CHECK_TURB_HISTORY": "1"})


def test_slab_front_end_restarts_from_the_file_of_the_whole_box(comm_emu_lib, tmp_path):
    """3 slabs resume at step 3 from the ghost-inclusive .h5 a 2-slab run wrote and end in the files of the uninterrupted
    single-domain run (the worker compares the restarted slabs with a restarted single-domain run; both are compared with the
    uninterrupted one here)"""
    import h5util
    if not h5util.available():
        pytest.skip("no loadable libhdf5 on this machine")
    base = "mhd_mri_3d"
    ov = "mesh.nx=6;mesh.ny=8;mesh.nz=12;MRI.amp=0.2;run.nstepmax=6;run.noutput=3;run.tend=1e9;output.outputVtk=no;output.outputHdf5=yes;output.ghostIncluded=yes"
    run_frontend(base, ov, 2, tmp_path / "a", tmp_path)
    first = tmp_path / "a" / "slabs"
    f3 = [f for f in os.listdir(first) if f.endswith("0000003.h5")][0]
    for sub in ("slabs", "single"):
        os.makedirs(tmp_path / "b" / sub)
        (tmp_path / "b" / sub / f3).write_bytes((first / f3).read_bytes())
    run_frontend(base, ov + ";run.restart=yes;run.restart_filename=%s" % f3, 3, tmp_path / "b", tmp_path)
    f6 = f3.replace("0000003", "0000006")
    da, aa = h5util.read(str(first / f6))
    db, ab = h5util.read(str(tmp_path / "b" / "slabs" / f6))
    assert aa == ab and all(np.array_equal(da[k], db[k]) for k in da)


GPU_CASES = [
    ("mhd_mri_3d", "mesh.nx=32;mesh.ny=48;mesh.nz=40", 4, 1),            # periodic z: grouped ncclSend / ncclRecv to itself
    ("orszag-tang3d", "mesh.nx=24;mesh.ny=24;mesh.nz=40", 3, 1),
    ("orszag-tang3d", "mesh.nx=24;mesh.ny=24;mesh.nz=40", 3, 0),         # serial schedule
    ("implode3d", "mesh.nx=32;mesh.ny=32;mesh.nz=32;hydro.riemannSolver=hllc", 4, 1),
    ("mhd_mri_3d", "mesh.nx=32;mesh.ny=48;mesh.nz=40", 4, 2),            # boundary-first: three launches of the LDS-tiled sweep per step
    ("orszag-tang3d", "mesh.nx=24;mesh.ny=24;mesh.nz=40", 3, 2),
]


@pytest.mark.gpu
@pytest.mark.parametrize("base,ov,nsteps,overlap", GPU_CASES, ids=["%s-%s" % (c[0], ("serial", "overlap", "boundary-first")[c[3]]) for c in GPU_CASES])
def test_rccl_driver_single_rank_on_gpu(base, ov, nsteps, overlap, gpu_lib, oracle, tmp_path_factory):
    """the RCCL transport with nranks = 1 on the 1-GPU box (a ring of one): product libraries, no torch in the data path"""
    cases = [(c[0], c[1], c[2], 1, c[3]) for c in GPU_CASES]
    run_batched("rccl1", cases, (base, ov, nsteps, 1, overlap), tmp_path_factory, env_extra={"COMM_DEVICE": "cuda:0", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})


@pytest.mark.gpu
@pytest.mark.parametrize("base,ov", [("mhd_mri_3d", "mesh.nx=32;mesh.ny=48;mesh.nz=40"), ("orszag-tang3d", "mesh.nx=24;mesh.ny=24;mesh.nz=40"),
                                     ("implode3d", "mesh.nx=32;mesh.ny=32;mesh.nz=32;hydro.riemannSolver=hllc")], ids=["mri", "ot3d", "implode3d"])
def test_overlapped_slab_steps_need_no_scan_kernel(base, ov, tmp_path):
    """the CFL scan of the overlapped slab schedule rides in the update kernels (RGPU_CORE_SCAN): after the first step the
    phase timers show no stand-alone scan, and the time steps equal those of the single-device run.  In a subprocess (RCCL)."""
    code = r'''
import os, sys
import numpy as np
sys.path.insert(0, %r)
from ramsesgpu_amd import comm as rcomm
from ramsesgpu_amd.solver import Solver, load_library
L = load_library(); CL = rcomm.load_comm_library()
ini = os.path.join(%r, "configs", %r + ".ini"); ov = %r
RING = %r != "implode3d"   # periodic z: the slab is its own neighbour and really exchanges its planes
run = rcomm.CommRun(ini, ov, 0, 1, rcomm.unique_id(CL), library=L, comm_library=CL, overlap=True, self_ring=RING)
assert (run.halo_bytes() > 0) == RING
run.init_simulation()
dts = [run.oneStepIntegration()]
run.solver.enable_timers(True); run.solver.reset_timers()
dts += [run.oneStepIntegration() for _ in range(4)]
tm = run.solver.timers(); run.close()
assert tm["dt"] == 0.0 and (tm["update"] > 0 or tm["sweep"] > 0), tm
p = L.params_from_ini(ini, ov); sv = Solver(p, L); ref = sv.start(L.init_condition(ini, ov, p), 5); sv.close()
assert list(dts) == list(ref), (dts, ref)
print("OK")
''' % (ROOT, ROOT, base, ov, base)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("schedule", [1, 2], ids=["overlap", "boundary-first"])
def test_config5_rank_geometry_through_the_rccl_driver(schedule, tmp_path):
    """BASELINE config 5 = 512 x 1024 x 512 MRI over 8 GPUs: every rank owns a 512 x 1024 x 64 slab (+ 3 ghost planes per side).
    That per-rank box through the product's slab driver (rgpu_comm, RCCL ring of one rank: the slab is its own z neighbour,
    205 MB sent per exchange on the halo stream, asserted) for 4 steps of the overlapped and of the boundary-first schedule (the
    default at this slab thickness): (i) every double and every dt equal to the
    single-device run of the same box, (ii) div B at round-off and mass conserved to round-off (size-independent
    properties: the oracle cannot run this size in seconds), (iii) RCCL itself reports 1 rank on the device the context uses."""
    code = r'''
import os, sys
import numpy as np
sys.path.insert(0, %r)
from ramsesgpu_amd import comm as rcomm
from ramsesgpu_amd.solver import Solver, load_library, interior
L = load_library(); CL = rcomm.load_comm_library()
ini = os.path.join(%r, "configs", "mhd_mri_3d.ini"); ov = "mesh.nx=512;mesh.ny=1024;mesh.nz=64"
run = rcomm.CommRun(ini, ov, 0, 1, rcomm.unique_id(CL), library=L, comm_library=CL, overlap=%d, self_ring=True)
assert run.halo_bytes() == 2 * 3 * 518 * 1030 * 8 * 8, run.halo_bytes()
info = run.info()
assert info["ranks"] == 1 and info["rank"] == 0 and info["device"] == 0 and info["transport"] == "rccl" and info["pci_bus_id"], info
run.init_simulation()
p = run.p
gw = p.ghostWidth
A0 = run.solver.getDataHost(0)
mass0 = float(A0[0, gw:-gw, gw:-gw, gw:-gw].sum(dtype=np.longdouble))
dts = [run.oneStepIntegration() for _ in range(4)]
xms = run.last_exchange_ms()
assert 0.0 < xms < 1000.0, xms   # the exchange really ran on the halo stream (time stamps around the grouped send / recv)
got = run.solver.getDataHost(run.nStep %% 2).copy()
run.close()
ps = L.params_from_ini(ini, ov); sv = Solver(ps, L); ref_dts = sv.start(L.init_condition(ini, ov, ps), 4); ref = sv.getDataHost(); sv.close()
assert list(dts) == list(ref_dts), (dts, ref_dts)
gi, ri = interior(got, p), interior(ref, ps)
nbad = int((gi != ri).sum())
assert nbad == 0, "%%d of %%d doubles differ from the single-device run" %% (nbad, ri.size)
bx, by, bz = got[5], got[6], got[7]
s = (slice(gw, -gw),) * 3
d = (bx[gw:-gw, gw:-gw, gw + 1:-gw + 1] - bx[s]) / p.dx + (by[gw:-gw, gw + 1:-gw + 1, gw:-gw] - by[s]) / p.dy + (bz[gw + 1:-gw + 1, gw:-gw, gw:-gw] - bz[s]) / p.dz
bscale = float(np.abs(bz[s]).max()) / min(p.dx, p.dy, p.dz)
assert float(np.abs(d).max()) < 1e-11 * bscale, (float(np.abs(d).max()), bscale)
mass1 = float(gi[0].sum(dtype=np.longdouble))
assert abs(mass1 - mass0) < 1e-12 * abs(mass0), (mass0, mass1)
assert np.isfinite(gi).all()
print("OK")
''' % (ROOT, ROOT, schedule)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=900)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("arith", ["exact", "contracted"])
def test_config5_whole_box_on_one_gpu_properties(arith):
    """BASELINE config 5, the WHOLE box: 512 x 1024 x 512 MRI (2.7e8 cells, 75 GB at 34 doubles per cell) on one MI355X through
    the product's slab driver (rgpu_comm, RCCL ring of one rank, overlapped schedule), 3 steps, both arithmetics.  The oracle
    cannot run this size, so size-independent properties: div B at round-off, mass conserved to round-off (shearing-box remap
    included), finite fields, a decreasing-or-equal CFL time step sequence of sane magnitude.  What remains untested of config 5
    is only the N > 1 RCCL exchange itself (no multi-GPU box is available to the builder; the driver's SCALE run covers it)."""
    code = r"""
import os, sys
import numpy as np
sys.path.insert(0, %r)
from ramsesgpu_amd import comm as rcomm
from ramsesgpu_amd.solver import Library, lib_path
arith = %r
L = Library(lib_path(arith)); assert L.arithmetic == arith
CL = rcomm.load_comm_library(rcomm.comm_lib_path(arith))
ini = os.path.join(%r, "configs", "mhd_mri_3d.ini"); ov = "mesh.nx=512;mesh.ny=1024;mesh.nz=512"
run = rcomm.CommRun(ini, ov, 0, 1, rcomm.unique_id(CL), library=L, comm_library=CL, overlap=True, self_ring=True)
assert run.halo_bytes() == 2 * 3 * 518 * 1030 * 8 * 8, run.halo_bytes()
info = run.info()
assert info["ranks"] == 1 and info["transport"] == "rccl", info
run.init_simulation()
p = run.p; gw = p.ghostWidth
assert (p.nx, p.ny, p.nz) == (512, 1024, 512)
def props(A):
    s = (slice(gw, -gw),) * 3
    mass = 0.0
    dmax = 0.0
    for k0 in range(gw, A.shape[1] - gw, 64):          # plane chunks: temporaries stay small next to the 17.7 GB array
        k1 = min(k0 + 64, A.shape[1] - gw)
        c = (slice(k0, k1), slice(gw, -gw), slice(gw, -gw))
        mass += float(A[0][c].sum(dtype=np.longdouble))
        bx, by, bz = A[5], A[6], A[7]
        d = (bx[k0:k1, gw:-gw, gw + 1:-gw + 1] - bx[c]) / p.dx + (by[k0:k1, gw + 1:-gw + 1, gw:-gw] - by[c]) / p.dy + (bz[k0 + 1:k1 + 1, gw:-gw, gw:-gw] - bz[c]) / p.dz
        dmax = max(dmax, float(np.abs(d).max()))
    return mass, dmax
A = run.solver.getDataHost(0)
mass0, d0 = props(A)
bscale = float(np.abs(A[7]).max()) / min(p.dx, p.dy, p.dz)
del A
dts = [run.oneStepIntegration() for _ in range(3)]
B = run.solver.getDataHost(run.nStep %% 2)
run.close()
assert np.isfinite(B).all()
mass1, d1 = props(B)
assert abs(mass1 - mass0) < 1e-13 * abs(mass0), (mass0, mass1)
assert d1 <= max(d0, 1e-13 * bscale) + 1e-12 * bscale, (d0, d1, bscale)
assert all(0 < dt < 1.0 for dt in dts) and max(dts) / min(dts) < 1.01, dts
print("config5 whole box, %%s arithmetic: dt %%s, mass drift %%.1e, div B %%.1e (scale %%.1e)" %% (arith, dts, abs(mass1 / mass0 - 1), d1, bscale))
print("OK")
""" % (ROOT, arith, ROOT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"); env.pop("RGPU_LIB", None); env.pop("RGPU_ARITH", None)
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=1500)
    print(out.stdout[-600:])
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-3000:]


@pytest.mark.gpu
def test_euler_hip_slabs_front_end_single_rank(gpu_lib, tmp_path):
    """euler_hip --slabs 1: rendezvous file, rgpuh_run_slabs, RCCL self ring -- same step count and dt log as the single-GPU run"""
    exe = os.path.join(ROOT, "ramsesgpu_amd", "euler_hip")
    ov = "mesh.nx=16;mesh.ny=24;mesh.nz=16;run.nstepmax=6;run.noutput=1;output.outputVtk=no;output.outputHdf5=no"
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    a = subprocess.run([exe, "--param", ini("mhd_mri_3d"), "--set", ov, "--slabs", "1", "--rendezvous", str(tmp_path / "rdv")],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=300)
    b = subprocess.run([exe, "--param", ini("mhd_mri_3d"), "--set", ov], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       universal_newlines=True, timeout=300)
    assert a.returncode == 0, a.stdout[-2000:]
    assert b.returncode == 0, b.stdout[-2000:]
    dts = lambda txt: re.findall(r"step=\s*(\d+) t=\s*([0-9.eE+-]+) dt=\s*([0-9.eE+-]+)", txt)
    da, db = dts(a.stdout), dts(b.stdout)
    assert "steps 6" in a.stdout and da, a.stdout[-1000:]
    # the front end binds LOCAL_RANK -> HIP device before the context exists and logs what RCCL itself reports
    assert re.search(r"rank 0/1 -> HIP device 0 \(PCI [0-9a-fA-F:.]+\), RCCL communicator of 1 rank", a.stdout), a.stdout[:1500]
    # the slab front end prints after each step, the single-GPU one before: compare the (t, dt) pairs they share
    assert set(x[1:] for x in da) & set(x[1:] for x in db), (da, db)
