"""The PRODUCT's C++ z-slab driver with N > 1 ranks on the real tiled HIP kernels -- on the ONE GPU of the test box.

RCCL refuses a second rank on a device, so until round 5 the matrix had an empty cell: C++ schedule x real kernels x a real
neighbour.  Here csrc/comm/rgpu_comm.cpp is compiled against the TEST-ONLY device-aware transport (tests/emu_dev/rg_transport.h)
and linked with the product library (librgpu.so / librgpu_fast.so): 2 and 3 rank processes share cuda:0; halo stream, event
ordering, the product's pack / unpack kernels and staging plan (csrc/hip/halo_pack.h), the in-place all-reduce of the 1/dt device
slots are the product's; only the wire (ncclSend / ncclRecv / ncclAllReduce) is replaced by pinned host buffers + gloo.

exact library      == the single-domain oracle, every double and every dt (small boxes); == the single-device run of the whole box
                      through the same library (bench-size slabs of 64 planes)
contracted library == its own single-device run, every double and every dt
and the N-independent fingerprint (dt sequence + order-independent state checksum, what bench.py prints) is the same for
single device, ring of one through RCCL, and the 2-rank run.

not gpu: the test-transport build of the driver compiles and exports the driver's symbols (hipcc cross-compiles here)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, ini
from test_comm_driver import OPEN_BC, run_batched, run_frontend, run_worker


def build_dev_comm(arith):
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    fast = arith != "exact"
    so = os.path.join(out_dir, "librgpu_comm_dev%s.so" % ("_fast" if fast else ""))
    csrc = os.path.join(ROOT, "ramsesgpu_amd", "csrc")
    src = os.path.join(csrc, "comm", "rgpu_comm.cpp")
    core = os.path.join(ROOT, "ramsesgpu_amd", "librgpu_fast.so" if fast else "librgpu.so")
    if not os.path.exists(core):
        import __graft_entry__
        __graft_entry__.build()
    deps = [src, os.path.join(ROOT, "tests", "emu_dev", "rg_transport.h"), os.path.join(csrc, "hip", "halo_pack.h"), os.path.join(csrc, "comm", "pack_plan.h"),
            os.path.join(csrc, "comm", "halo_ops.h"), os.path.join(ROOT, "include", "rgpu_comm.h"), os.path.join(ROOT, "include", "rgpu.h"), core]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wl,-Bsymbolic-functions",
                               "-I", os.path.join(ROOT, "tests", "emu_dev"), "-I", os.path.join(csrc, "hip"), "-x", "hip", src, "-x", "none",
                               "-L", os.path.join(ROOT, "ramsesgpu_amd"), "-lrgpu_fast" if fast else "-lrgpu", "-Wl,-rpath,$ORIGIN/../../ramsesgpu_amd", "-o", so])
    return so


@pytest.fixture(scope="session")
def dev_comm_exact():
    return build_dev_comm("exact")


@pytest.fixture(scope="session")
def dev_comm_contracted():
    return build_dev_comm("contracted")


def test_device_transport_build_exports_the_driver(dev_comm_exact):
    from ramsesgpu_amd import comm as rcomm
    out = subprocess.check_output(["nm", "-D", "--defined-only", dev_comm_exact], universal_newlines=True)
    exported = set(l.split()[-1] for l in out.splitlines() if l.strip())
    for sym in rcomm.DECLARED_SYMBOLS + ["rgpu_comm_test_set_callbacks", "rgpu_comm_test_stats"]:
        assert sym in exported, sym
    # the test build must not carry RCCL
    needed = subprocess.check_output(["readelf", "-d", dev_comm_exact], universal_newlines=True)
    assert "librccl" not in needed and "librgpu.so" in needed


MRI = "MHD.omega0=0.02"
# (ini, overrides, steps, world, schedule)
SMALL = [
    # MRI shearing box, periodic ring
    ("mhd_mri_3d", "mesh.nx=16;mesh.ny=24;mesh.nz=40;" + MRI, 4, 2, 1),      # N = 2: both neighbours are the SAME peer (one packed message each way)
    ("mhd_mri_3d", "mesh.nx=16;mesh.ny=24;mesh.nz=40;" + MRI, 4, 2, 2),
    ("mhd_mri_3d", "mesh.nx=16;mesh.ny=24;mesh.nz=40;" + MRI, 3, 2, 0),
    ("mhd_mri_3d", "mesh.nx=16;mesh.ny=16;mesh.nz=60", 4, 3, 1),             # N = 3: two distinct neighbours
    ("mhd_mri_3d", "mesh.nx=16;mesh.ny=16;mesh.nz=60", 4, 3, 2),
    ("mhd_mri_3d", "mesh.nx=16;mesh.ny=16;mesh.nz=60", 3, 3, 0),
    ("mhd_mri_3d", "mesh.nx=16;mesh.ny=16;mesh.nz=12;" + MRI, 4, 2, 1),      # slabs of 6 = 2 gw planes: no inner planes
    ("mhd_mri_3d", "mesh.nx=16;mesh.ny=16;mesh.nz=12", 3, 3, 1),             # slabs of 4 < 2 gw planes
    ("mhd_mri_3d", "mesh.nx=16;mesh.ny=16;mesh.nz=12", 3, 3, 2),
    # implode: Dirichlet end slabs (hydro: the sweep is the whole step)
    ("implode3d", "mesh.nx=24;mesh.ny=24;mesh.nz=32;hydro.riemannSolver=hllc", 4, 2, 1),
    ("implode3d", "mesh.nx=24;mesh.ny=24;mesh.nz=36;hydro.riemannSolver=hllc", 4, 3, 1),
    ("implode3d", "mesh.nx=24;mesh.ny=24;mesh.nz=36", 3, 3, 0),
    ("implode3d", "mesh.nx=16;mesh.ny=16;mesh.nz=8;hydro.riemannSolver=hllc", 3, 2, 1),   # slabs of 4 = 2 gw planes (hydro gw = 2)
    # plain 3D MHD, periodic and open
    ("orszag-tang3d", "mesh.nx=16;mesh.ny=16;mesh.nz=32", 4, 2, 1),
    ("orszag-tang3d", "mesh.nx=16;mesh.ny=16;mesh.nz=48", 4, 3, 2),
    ("orszag-tang3d", "mesh.nx=16;mesh.ny=16;mesh.nz=32" + OPEN_BC, 4, 2, 1),
    ("orszag-tang3d", "mesh.nx=16;mesh.ny=16;mesh.nz=48" + OPEN_BC, 3, 3, 2),
]
SCHED = ("serial", "overlap", "boundary-first")
ENV = {"HSA_ENABLE_IPC_MODE_LEGACY": "0", "COMM_DEVICE": "cuda-staged:0"}


@pytest.mark.gpu
@pytest.mark.parametrize("base,ov,nsteps,world,overlap", SMALL, ids=["%s-%d-x%d-%s" % (c[0], n, c[3], SCHED[c[4]]) for n, c in enumerate(SMALL)])
def test_exact_ranks_on_one_gpu_equal_the_oracle(base, ov, nsteps, world, overlap, dev_comm_exact, gpu_lib, oracle, tmp_path_factory):
    run_batched("small_exact", SMALL, (base, ov, nsteps, world, overlap), tmp_path_factory, env_extra=dict(ENV, COMM_ARITH="exact"))


@pytest.mark.gpu
def test_exact_ranks_in_place_exchange(dev_comm_exact, gpu_lib, oracle, tmp_path):
    """RGPU_COMM_PACK=0: one message per chunk (8 variables x 2 faces), the round 1-3 wire format"""
    run_worker("mhd_mri_3d", "mesh.nx=16;mesh.ny=24;mesh.nz=40;" + MRI, 3, 2, 1, tmp_path, env_extra=dict(ENV, COMM_ARITH="exact", RGPU_COMM_PACK="0"), timeout=600)


RUN_STEPS = [SMALL[0], SMALL[1], SMALL[3], SMALL[4], SMALL[6], SMALL[7], SMALL[9], SMALL[10], SMALL[13], SMALL[14], SMALL[15], SMALL[16]]


@pytest.mark.gpu
@pytest.mark.parametrize("base,ov,nsteps,world,overlap", RUN_STEPS, ids=["%s-%d-x%d-%s" % (c[0], n, c[3], SCHED[c[4]]) for n, c in enumerate(RUN_STEPS)])
def test_exact_ranks_run_steps_with_the_time_step_on_the_device(base, ov, nsteps, world, overlap, dev_comm_exact, gpu_lib, oracle, tmp_path_factory):
    """rgpu_comm_run_steps on the real kernels with a real neighbour: the 1/dt slots all-reduced in place, the clock kernel, the sweeps /
    update / shear remap / fused ghost fill reading the record; one plain step then a batch of 5, and an end time inside a batch"""
    run_batched("run_steps_exact", RUN_STEPS, (base, ov, nsteps, world, overlap), tmp_path_factory, env_extra=dict(ENV, COMM_ARITH="exact", COMM_RUN_STEPS="1"), nsteps=6)


# the rank counts of the scaling run (SCALE: N = 2, 4, 8): every rank a process of its own on the one GPU
MANY = [("mhd_mri_3d", "mesh.nx=16;mesh.ny=24;mesh.nz=64;" + MRI, 4, 4, 1),
        ("mhd_mri_3d", "mesh.nx=16;mesh.ny=24;mesh.nz=128;" + MRI, 4, 8, 1),          # 8 slabs of 16 planes: the shape of BASELINE config 5
        ("mhd_mri_3d", "mesh.nx=16;mesh.ny=24;mesh.nz=128;" + MRI, 4, 8, 2),
        ("implode3d", "mesh.nx=24;mesh.ny=24;mesh.nz=64;hydro.riemannSolver=hllc", 4, 8, 1)]


@pytest.mark.gpu
@pytest.mark.parametrize("base,ov,nsteps,world,overlap", MANY, ids=["%s-x%d-%s" % (c[0], c[3], SCHED[c[4]]) for c in MANY])
def test_exact_four_and_eight_ranks_on_one_gpu(base, ov, nsteps, world, overlap, dev_comm_exact, gpu_lib, oracle, tmp_path_factory):
    """4 and 8 rank processes (the rank counts of the scaling run) through the batched loop: == the single-domain oracle, dt sequence,
    fingerprint and an end time inside a batch included"""
    run_batched("many_exact", MANY, (base, ov, nsteps, world, overlap), tmp_path_factory, env_extra=dict(ENV, COMM_ARITH="exact", COMM_RUN_STEPS="1"), nsteps=6)


CONTRACTED = [SMALL[0], SMALL[1], SMALL[3], SMALL[4], SMALL[6], SMALL[9], SMALL[14]]


@pytest.mark.gpu
@pytest.mark.parametrize("base,ov,nsteps,world,overlap", CONTRACTED, ids=["%s-%d-x%d-%s" % (c[0], n, c[3], SCHED[c[4]]) for n, c in enumerate(CONTRACTED)])
def test_contracted_ranks_on_one_gpu_equal_their_single_device_run(base, ov, nsteps, world, overlap, dev_comm_contracted, gpu_contracted_lib, tmp_path_factory):
    run_batched("small_contracted", CONTRACTED, (base, ov, nsteps, world, overlap), tmp_path_factory, env_extra=dict(ENV, COMM_ARITH="contracted", COMM_CHECK="single"))


# bench geometry: the 512 x 512 cross-section of the headline box in slabs of 64 planes (the N = 8 per-rank slab), 2 and 3 of them
BIG = [("mhd_mri_3d", "mesh.nx=512;mesh.ny=512;mesh.nz=128", 3, 2, 1, "exact"),
       ("mhd_mri_3d", "mesh.nx=512;mesh.ny=512;mesh.nz=128", 3, 2, 2, "exact"),
       ("mhd_mri_3d", "mesh.nx=512;mesh.ny=512;mesh.nz=192", 3, 3, 1, "contracted"),
       ("mhd_mri_3d", "mesh.nx=512;mesh.ny=512;mesh.nz=192", 3, 3, 2, "exact")]


@pytest.mark.gpu
@pytest.mark.parametrize("base,ov,nsteps,world,overlap,arith", BIG, ids=["x%d-%s-%s" % (c[3], SCHED[c[4]], c[5]) for c in BIG])
def test_bench_size_slabs_of_64_planes(base, ov, nsteps, world, overlap, arith, dev_comm_exact, dev_comm_contracted, gpu_lib, gpu_contracted_lib, tmp_path):
    """2 x and 3 x (512 x 512 x 64) through the product's schedule on the tiled kernels == the single-device run of the whole box"""
    run_worker(base, ov, nsteps, world, overlap, tmp_path, env_extra=dict(ENV, COMM_ARITH=arith, COMM_CHECK="single"), timeout=1200)


def build_fail_shim():
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so, src = os.path.join(out_dir, "librgpu_fail_shim.so"), os.path.join(ROOT, "tests", "shim", "fail_shim.cpp")
    if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", src, "-ldl", "-o", so])
    return so


def test_fail_shim_builds_and_exports_the_step_piece():
    out = subprocess.check_output(["nm", "-D", "--defined-only", build_fail_shim()], universal_newlines=True)
    assert " T rgpu_step_fill_planes_pair" in out and " T rgpu_test_fail_after" in out


POISON_GPU = [
    # (ini, overrides, world, failing rank, failing step, loop): a step piece fails on ONE rank of the device backend, where the host does
    # not see the records of a batch until it ends -- every rank must come back with an error, nobody may wait in a collective
    ("orszag-tang3d", "mesh.nx=16;mesh.ny=16;mesh.nz=32", 2, 1, 0, "host"),      # the reference's per-step loop
    ("orszag-tang3d", "mesh.nx=16;mesh.ny=16;mesh.nz=32", 2, 0, 2, "batch"),     # first step of a batch of three: the failed rank pairs the queued steps
    ("mhd_mri_3d", "mesh.nx=16;mesh.ny=16;mesh.nz=60;" + MRI, 3, 1, 1, "batch"),
    ("orszag-tang3d", "mesh.nx=16;mesh.ny=16;mesh.nz=32", 2, 1, 1, "single"),    # last (only) step of a batch: told through the next call's all-reduce,
    ("mhd_mri_3d", "mesh.nx=16;mesh.ny=16;mesh.nz=60;" + MRI, 3, 0, 2, "single"),    # whose first record the healthy ranks check on the host
    ("implode3d", "mesh.nx=24;mesh.ny=24;mesh.nz=32;hydro.riemannSolver=hllc", 2, 1, 1, "batch"),
]


@pytest.mark.gpu
@pytest.mark.parametrize("base,ov,world,fail_rank,fail_step,loop", POISON_GPU, ids=["%s-x%d-r%d-s%d-%s" % (c[0], c[2], c[3], c[4], c[5]) for c in POISON_GPU])
def test_a_failed_step_piece_on_one_rank_reaches_every_rank_on_the_device(base, ov, world, fail_rank, fail_step, loop, dev_comm_exact, gpu_lib, tmp_path):
    """round-5 ADVICE (medium), on the backend it was about: the product's slab driver + tiled kernels, rank processes on one GPU (test
    wire), a step piece failing on one rank (tests/shim/fail_shim.cpp between librgpu_comm and librgpu).  The failing rank posts the
    exchange its neighbours wait for and poisons the 1/dt all-reduce; healthy ranks that have queued a batch find stop = 3 in its records
    (or in the host-checked first record of their next batch) and return "1/dt is not finite"."""
    from ramsesgpu_amd.solver import lib_path
    from test_comm_driver import free_port
    import sys
    out = str(tmp_path / "result.txt")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "comm_worker.py"), "--poison", base, ov, str(fail_rank), str(fail_step), out]
    env = dict(os.environ, OMP_NUM_THREADS="1", COMM_OVERLAP="1", COMM_ARITH="exact", LD_PRELOAD=build_fail_shim(), RGPU_SHIM_REAL=lib_path("exact"), **ENV)
    if loop != "host":
        env["POISON_RUN_STEPS"] = loop
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=300)
    assert res.returncode == 0, res.stdout[-3000:]
    assert open(out).read().strip() == "OK", open(out).read()


FRONTEND_GPU = [
    # whole-box HDF5 file written rank after rank + XDMF index, and per-rank .vti pieces + .pvti index, from 2 and 3 rank processes
    ("mhd_mri_3d", "mesh.nx=16;mesh.ny=24;mesh.nz=40;MRI.amp=0.2;run.nstepmax=6;run.noutput=3;run.tend=1e9;output.outputVtk=yes;output.outputHdf5=yes", 2),
    ("mhd_mri_3d", "mesh.nx=16;mesh.ny=16;mesh.nz=36;MRI.amp=0.2;run.nstepmax=4;run.noutput=2;run.tend=1e9;output.outputVtk=yes;output.outputHdf5=yes;output.ghostIncluded=yes", 3),
    ("implode3d", "mesh.nx=24;mesh.ny=24;mesh.nz=32;hydro.riemannSolver=hllc;run.nstepmax=4;run.noutput=2;run.tend=1e9;output.outputVtk=yes;output.outputHdf5=yes", 2),
]


@pytest.mark.gpu
@pytest.mark.parametrize("base,ov,world", FRONTEND_GPU, ids=["%s-x%d-%d" % (c[0], c[2], n) for n, c in enumerate(FRONTEND_GPU)])
def test_slab_front_end_on_the_device_writes_the_single_domain_files(base, ov, world, dev_comm_exact, gpu_lib, tmp_path):
    """SURVEY 8(f3), multi-GPU output on the device path (HydroRunBaseMpi.cpp:4167-4790 pvti / pieces, :4845-6700 parallel HDF5): the
    product's run loop rgpuh_run_slabs with outputHdf5 = outputVtk = yes, every rank a process on the one GPU, the product's tiled
    kernels and slab driver (test wire).  The .h5 files equal the single-domain run's dataset by dataset and attribute by
    attribute (same library, same GPU), the XDMF index is the same text, the .pvti names the pieces with the reference's extents
    and every piece holds the doubles of the single-domain .vti (the CPU emulation runs the same worker: test_comm_driver.py)."""
    import h5util
    if not h5util.available():
        pytest.skip("no loadable libhdf5 on this machine")
    run_frontend(base, ov, world, tmp_path / "run", tmp_path, timeout=600, env_extra=dict(ENV, COMM_ARITH="exact"))


@pytest.mark.gpu
def test_fingerprint_is_independent_of_the_rank_count(dev_comm_exact, gpu_lib, oracle, tmp_path):
    """dt-sequence hash + state checksum (bench.py's config.fingerprint): single device == ring of one through RCCL == two ranks
    through the staged transport == three ranks, all on the exact library; and the bench helper computes the same value"""
    import hashlib
    import sys
    sys.path.insert(0, ROOT)
    import bench
    from ramsesgpu_amd.solver import Solver
    base, ov, nsteps = "mhd_mri_3d", "mesh.nx=16;mesh.ny=24;mesh.nz=48;" + MRI, 4
    p = gpu_lib.params_from_ini(ini(base), ov)
    one = Solver(p, gpu_lib)
    one.upload(gpu_lib.init_condition(ini(base), ov, p), both=False)
    one.make_all_boundaries(0, 0.0, 0.0)
    dts = [one.oneStepIntegration() for _ in range(nsteps)]
    fp_one = one.state_checksum(one.nStep % 2)
    gw = p.ghostWidth
    host = np.ascontiguousarray(one.getDataHost(one.nStep % 2)[:, gw:-gw, gw:-gw, gw:-gw])
    one.close()
    assert fp_one == int(host.view(np.uint64).sum(dtype=np.uint64))       # the device kernel == numpy on the downloaded state
    want = "%016x" % fp_one
    fps = {}
    for world, env in ((1, {"COMM_DEVICE": "cuda:0", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}), (2, dict(ENV, COMM_ARITH="exact")), (3, dict(ENV, COMM_ARITH="exact"))):
        d = tmp_path / ("w%d" % world)
        d.mkdir()
        fps[world] = run_worker(base, ov, nsteps, world, 1, d, env_extra=env, timeout=600)[0]
    assert fps == {1: want, 2: want, 3: want}, (fps, want)
    rec = bench.fingerprint_record(dts, [fp_one])
    assert rec["state_sum_u64"] == want and rec["dt_sha256"] == hashlib.sha256(np.asarray(dts, dtype="<f8").tobytes()).hexdigest() and rec["steps"] == nsteps
