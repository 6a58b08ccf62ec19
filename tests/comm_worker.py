"""Worker of tests/test_comm_driver.py: one rank of a z-slab run through the C++ driver (include/rgpu_comm.h).

CPU (default): tests/_build/librgpu_comm_emu.so = csrc/comm/rgpu_comm.cpp compiled against the TEST-ONLY callback transport
(tests/emu/rg_transport.h) and the emulation backend; the callbacks registered here move the ghost planes and reduce
1/dt with torch.distributed / gloo.  The schedule, the op lists and the dt logic under test are the product's C++.
COMM_DEVICE=cuda:N: the product libraries (HIP + RCCL); torch.distributed only carries the 128-byte unique id.
Result == the single-domain oracle, bit for bit."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle_api import Oracle  # noqa: E402
from ramsesgpu_amd import _capi  # noqa: E402
from ramsesgpu_amd import comm as rcomm  # noqa: E402
from ramsesgpu_amd.solver import Library, interior  # noqa: E402


class P2P(C.Structure):
    _fields_ = [("ptr", C.POINTER(C.c_double)), ("count", C.c_size_t), ("peer", C.c_int), ("send", C.c_int)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(P2P), C.c_int)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int, C.c_int)


def _exchange(ops, nops):
    """all sends / receives of one exchange as one gloo batch; tags pair the n-th send to a peer with its n-th receive"""
    try:
        batch, nsend, nrecv = [], {}, {}
        for i in range(nops):
            o = ops[i]
            t = torch.from_numpy(np.ctypeslib.as_array(o.ptr, shape=(o.count,)))
            if o.send:
                tag = nsend.get(o.peer, 0); nsend[o.peer] = tag + 1
                batch.append(dist.P2POp(dist.isend, t, o.peer, tag=tag))
            else:
                tag = nrecv.get(o.peer, 0); nrecv[o.peer] = tag + 1
                batch.append(dist.P2POp(dist.irecv, t, o.peer, tag=tag))
        for w in dist.batch_isend_irecv(batch):
            w.wait()
        return 0
    except Exception as e:  # noqa: BLE001
        sys.stderr.write("exchange callback: %r\n" % (e,))
        return 1


def _allreduce(data, n, op):
    try:
        t = torch.from_numpy(np.ctypeslib.as_array(data, shape=(n,)))
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 0 else dist.ReduceOp.SUM)
        return 0
    except Exception as e:  # noqa: BLE001
        sys.stderr.write("allreduce callback: %r\n" % (e,))
        return 1


def load_libs(device):
    """the libraries of one rank.  "cpu": the emulation builds (host loops, callback transport).  "cuda-staged:N": N rank processes on
    ONE GPU through the PRODUCT library (real tiled HIP kernels) and the product's slab driver compiled against the TEST-ONLY
    device-aware transport (tests/emu_dev/rg_transport.h): the planes are packed by the product's kernels, staged through pinned
    memory and carried by gloo (RCCL refuses two ranks on one device).  "cuda:N": the product libraries (HIP + RCCL).
    Returns (library, comm library, callbacks to keep alive)."""
    key = (device, os.environ.get("COMM_ARITH", "exact"))
    if key in _LIBS:      # (batch mode: one process, many cases)
        return _LIBS[key]
    _LIBS[key] = out = _load_libs(device)
    return out


_LIBS = {}


def _load_libs(device):
    keep = []
    if device == "cpu":
        lib = Library(os.path.join(ROOT, "tests", "_build", "librgpu_emu.so"))
        CL = rcomm.load_comm_library(os.path.join(ROOT, "tests", "_build", "librgpu_comm_emu.so"))
        keep = [EXCHANGE_FN(_exchange), ALLREDUCE_FN(_allreduce)]
        CL.rgpu_comm_test_set_callbacks(keep[0], keep[1])
        assert b"test" in CL.rgpu_comm_transport_name()
    elif device.startswith("cuda-staged"):
        from ramsesgpu_amd.solver import lib_path
        arith = os.environ.get("COMM_ARITH", "exact")
        lib = Library(lib_path(arith))
        CL = rcomm.load_comm_library(os.path.join(ROOT, "tests", "_build", "librgpu_comm_dev%s.so" % ("" if arith == "exact" else "_fast")))
        CL.rgpu_comm_set_device(int(device.split(":")[1]) if ":" in device else 0)
        keep = [EXCHANGE_FN(_exchange), ALLREDUCE_FN(_allreduce)]
        CL.rgpu_comm_test_set_callbacks(keep[0], keep[1])
        assert b"device-staged" in CL.rgpu_comm_transport_name() and lib.arithmetic == arith and "hip" in lib.backend
    else:
        from ramsesgpu_amd.solver import load_library
        lib = load_library()
        CL = rcomm.load_comm_library()
        CL.rgpu_comm_set_device(int(device.split(":")[1]) if ":" in device else 0)
        assert CL.rgpu_comm_transport_name() == b"rccl"
    return lib, CL, keep


def check_pvti(lib, ini, ov, slabs, single, world):
    """per-rank .vti pieces + .pvti index of the slab run (HydroRunBaseMpi::outputVtk) against the single-domain .vti files: every
    piece holds its own planes (ranks > 0 one plane more, below: the format's overlap), the index names them with those extents"""
    import re
    from vtiutil import read_vti
    p = lib.params_from_ini(ini, ov)
    nx, ny, nzg = p.nx, p.ny, p.nz
    nzl = nzg // world
    steps = sorted(f for f in os.listdir(single) if f.endswith(".vti"))
    if len(steps) < 2:
        return False, "single-domain run wrote %d .vti files" % len(steps)
    for f in steps:
        num = re.search(r"_(\d{7})\.vti$", f).group(1)
        prefix = f[:-len("_%s.vti" % num)]
        ref, _ = read_vti(os.path.join(single, f))
        index = os.path.join(slabs, "%s_time%s.pvti" % (prefix, num))
        if not os.path.exists(index):
            return False, "no %s" % index
        text = open(index).read()
        if '<PImageData WholeExtent="0 %d 0 %d 0 %d" GhostLevel="0" Origin="0 0 0" Spacing="1 1 1">' % (nx - 1, ny - 1, nzg - 1) not in text:
            return False, "pvti header: %s" % text[:300]
        for r in range(world):
            name = "%s_time%s_mpi%05d.vti" % (prefix, num, r)
            zlo, zhi = (0, nzl - 1) if r == 0 else (r * nzl - 1, r * nzl + nzl - 1)
            if ' <Piece Extent="0 %d 0 %d %d %d " Source="%s"/>' % (nx - 1, ny - 1, zlo, zhi, name) not in text:
                return False, "pvti piece %d: %s" % (r, text)
            got, ext = read_vti(os.path.join(slabs, name))
            if ext != [0, nx - 1, 0, ny - 1, zlo, zhi] or sorted(got) != sorted(ref):
                return False, "%s: extent %s, arrays %s" % (name, ext, sorted(got))
            for k in ref:
                if not np.array_equal(got[k], ref[k][zlo:zhi + 1]):
                    return False, "%s: %s differs from the single-domain file in %d doubles" % (name, k, int((got[k] != ref[k][zlo:zhi + 1]).sum()))
    return True, ""


def check_turbulence_history(lib, ini, ov, slabs, single, world):
    """history file of a z-slab turbulence run = the row of HydroRunBaseMpi::history_mhd_turbulence, its quirks included (sum of
    the ranks' |mean B|; divB, helicity and mean_rhov of rank 0 alone), evaluated here with numpy on the single-domain .h5 states"""
    import h5util
    p = lib.params_from_ini(ini, ov)
    nzl = p.nz // world
    rows = [l.split() for l in open([os.path.join(slabs, f) for f in os.listdir(slabs) if f.endswith("history.txt")][0]) if not l.startswith("#")]
    head = [l for l in open([os.path.join(slabs, f) for f in os.listdir(slabs) if f.endswith("history.txt")][0]) if l.startswith("# totalTime")]
    if not head or len(head[0].split()) != 17:
        return False, "history header %r" % head
    files = sorted(f for f in os.listdir(single) if f.endswith(".h5"))
    if len(rows) < 2 or len(files) < len(rows):
        return False, "%d history rows, %d .h5 files" % (len(rows), len(files))
    dTau = p.dx * p.dy * p.dz / (p.xMax - p.xMin) / (p.yMax - p.yMin) / (p.zMax - p.zMin)
    for n, row in enumerate(rows):
        d, a = h5util.read(os.path.join(single, files[n]))
        if abs(float(row[0]) - a["total time"]) > 1e-5 * max(abs(a["total time"]), 1e-30) + 1e-30:
            return False, "row %d is for t = %s, file %s for t = %r" % (n, row[0], files[n], a["total time"])
        gw = p.ghostWidth
        U = {k: d[k] for k in d}     # ghost-inclusive arrays [z, y, x] ([output] ghostIncluded=yes)
        I = (slice(gw, -gw),) * 3
        def slab(x, r):
            return x[I][r * nzl:(r + 1) * nzl]
        tot = dict(mass=0.0, v2=0.0, ek=0.0, em=0.0, bn=0.0, b=np.zeros(3))
        loc = None
        for r in range(world):
            rho = slab(U["density"], r)
            m = [slab(U["momentum_" + c], r) for c in "xyz"]
            b = [slab(U["magnetic_field_" + c], r) for c in "xyz"]
            mb = np.array([x.sum() for x in b]) * dTau
            tot["mass"] += rho.sum() * dTau
            tot["v2"] += sum(((x / rho) ** 2).sum() for x in m) * dTau
            tot["ek"] += sum((x * x / rho).sum() for x in m) * dTau
            tot["em"] += sum((x * x).sum() for x in b) * dTau
            tot["bn"] += float(np.sqrt((mb ** 2).sum()))
            tot["b"] += mb
            if r == 0:
                bx, by, bz = (U["magnetic_field_" + c] for c in "xyz")
                k0, k1 = gw, gw + nzl
                div = ((bx[k0:k1, gw:-gw, gw + 1:-gw + 1] - bx[k0:k1, gw:-gw, gw:-gw]) / p.dx + (by[k0:k1, gw + 1:-gw + 1, gw:-gw] - by[k0:k1, gw:-gw, gw:-gw]) / p.dy +
                       (bz[k0 + 1:k1 + 1, gw:-gw, gw:-gw] - bz[k0:k1, gw:-gw, gw:-gw]) / p.dz).sum()
                loc = dict(div=div, hel=sum((x * y / np.sqrt(rho)).sum() for x, y in zip(m, b)) * dTau, rv=[x.sum() * dTau for x in m])
        exp = [tot["mass"], loc["div"], tot["ek"], tot["em"], loc["hel"], tot["bn"], tot["b"][0], tot["b"][1], tot["b"][2], loc["rv"][0], loc["rv"][1],
               loc["rv"][2], np.sqrt(tot["v2"]) / p.cIso, np.sqrt(tot["v2"]) / (tot["bn"] / np.sqrt(4 * np.pi * tot["mass"]))]
        got = [float(x) for x in row[2:]]
        for q, (g, e) in enumerate(zip(got, exp)):
            noise = 1e-9 if q == 1 else 1e-14 if q in (4, 9, 10, 11) else 0.0   # div B, helicity, mean momentum: round-off noise around 0
            if abs(g - e) > 2e-5 * abs(e) + noise:    # 6 printed digits
                return False, "history row %d column %d: %r, expected %r" % (n, q + 2, g, e)
    return True, ""


def frontend():
    """--frontend base overrides outdir resultfile: the z-slab FRONT END (rgpuh_run_slabs: run loop, HDF5 outputs of the whole box
    written slab after slab, restart) on the emulation libraries; rank 0 then runs the single-domain front end (rgpuh_run) with
    the same settings and requires the .h5 files of both runs to hold the same datasets and attributes."""
    import h5util
    base, ov, outdir, out = sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    # COMM_BREAK_HDF5_ON_RANK=r: rank r alone cannot load libhdf5 -- every rank must come back with an error (no hang)
    broken = os.environ.get("COMM_BREAK_HDF5_ON_RANK")
    if broken is not None and int(broken) == rank:
        os.environ["RGPU_HDF5_LIB"] = "/nonexistent/libhdf5.so"
    device = os.environ.get("COMM_DEVICE", "cpu")   # "cuda-staged:0": the product's kernels and driver, every rank a process on the one GPU
    lib, CL, keep = load_libs(device)
    dev_index = int(device.split(":")[1]) if ":" in device else 0
    ids = [rcomm.unique_id(CL) if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    ini = os.path.join(ROOT, "configs", base + ".ini")
    slabs, single = os.path.join(outdir, "slabs"), os.path.join(outdir, "single")
    if rank == 0:
        os.makedirs(slabs, exist_ok=True); os.makedirs(single, exist_ok=True)
    dist.barrier()
    os.chdir(slabs)
    err = C.create_string_buffer(512); mc = C.c_double(0)
    CL.rgpuh_run_slabs.restype = C.c_int
    CL.rgpuh_run_slabs.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_double), C.c_char_p, C.c_int]
    n = CL.rgpuh_run_slabs(ini.encode(), (ov + ";output.outputDir=%s" % slabs).encode(), rank, world, dev_index, ids[0], C.byref(mc), err, 512)
    ok, msg = n >= 0, err.value.decode()
    if broken is not None:   # expected outcome: EVERY rank failed, together, with a message that names the count
        flags = [None] * world
        dist.all_gather_object(flags, (n < 0, msg))
        good = all(f[0] for f in flags) and all("ranks failed" in f[1] for f in flags) and "libhdf5" in flags[int(broken)][1]
        if rank == 0:
            with open(out, "w") as f:
                f.write("OK\n" if good else "FAILED %r\n" % (flags,))
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(0 if good else 1)
    dist.barrier()
    if rank == 0 and ok:
        os.chdir(single)
        m = lib.lib.rgpuh_run(ini.encode(), (ov.replace(slabs, single) + ";output.outputDir=%s" % single).encode(), C.byref(mc), err, 512)
        ok, msg = m == n, "steps %d vs %d %s" % (n, m, err.value.decode())
        files = sorted(f for f in os.listdir(single) if f.endswith(".h5"))
        want_h5 = "output.outputHdf5=yes" in ov
        ok = ok and (len(files) >= 2 or not want_h5) and files == sorted(f for f in os.listdir(slabs) if f.endswith(".h5"))
        for f in files if ok else []:
            da, aa = h5util.read(os.path.join(slabs, f)); db, ab = h5util.read(os.path.join(single, f))
            same = aa == ab and sorted(da) == sorted(db) and all(np.array_equal(da[k], db[k]) for k in db)
            if not same:
                ok, msg = False, "%s differs: attrs %s / %s, %s" % (f, aa, ab, {k: int((da[k] != db[k]).sum()) for k in db if da[k].shape == db[k].shape})
        turb = bool(os.environ.get("COMM_CHECK_TURB_HISTORY"))   # (the MPI classes' turbulence row has other columns: checked below)
        for f in [f for f in os.listdir(single) if f.endswith("history.txt")] if ok and not turb else []:   # global sums: round-off agreement, 6 printed digits
            ra = [l.split() for l in open(os.path.join(slabs, f)) if not l.startswith("#")]
            rb = [l.split() for l in open(os.path.join(single, f)) if not l.startswith("#")]
            same = len(ra) == len(rb) and len(rb) >= 2 and all(
                len(x) == len(y) and all(abs(float(u) - float(v)) <= 5e-6 * max(abs(float(v)), 1e-300) + 1e-14 for u, v in zip(x, y)) for x, y in zip(ra, rb))
            if not same:
                ok, msg = False, "history differs: %r / %r" % (ra, rb)
        xa = [f for f in os.listdir(slabs) if f.endswith(".xmf")]
        if files:
            ok = ok and len(xa) == 1 and open(os.path.join(slabs, xa[0])).read() == open(os.path.join(single, xa[0])).read()
    if rank == 0 and ok and "output.outputVtk=yes" in ov:
        ok, msg = check_pvti(lib, ini, ov, slabs, single, world)
    if rank == 0 and ok and os.environ.get("COMM_CHECK_TURB_HISTORY"):
        ok, msg = check_turbulence_history(lib, ini, ov, slabs, single, world)
    if rank == 0:
        with open(out, "w") as f:
            f.write("OK\n" if ok else "FAILED %s\n" % msg)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


def poison():
    """--poison base overrides fail_rank fail_step resultfile: rank `fail_rank` gets a launch error in the first step piece of step
    `fail_step` (emulation backend: rgpu_emu_fail_launch_after).  It must still post the halo exchange its neighbours wait for
    and tell them through the next 1/dt all-reduce -- whose size must be the one the healthy ranks use, whatever state the failing
    rank is in (step 0: its last all-reduce followed a full scan, theirs follows a fused one).  Expected: EVERY rank comes back
    with an error, the failing one at `fail_step`, the others one step later; nobody hangs (the test's timeout would tell)."""
    base, ov, frank, fstep, out = sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    device = os.environ.get("COMM_DEVICE", "cpu")
    lib, CL, keep = load_libs(device)
    if device == "cpu":
        fail_next = lambda: lib.lib.rgpu_emu_fail_launch_after(1)
    else:
        # the device backend: the product libraries have no failure injection; tests/shim/fail_shim.cpp, preloaded into this process,
        # stands between the driver and rgpu_step_fill_planes_pair (one call per step of the overlapped schedules)
        shim = C.CDLL(os.environ["LD_PRELOAD"].split(":")[0])
        shim.rgpu_test_fail_after.argtypes = [C.c_int]
        fail_next = lambda: shim.rgpu_test_fail_after(1)
    ids = [rcomm.unique_id(CL) if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    ini = os.path.join(ROOT, "configs", base + ".ini")
    run = rcomm.CommRun(ini, ov, rank, world, ids[0], library=lib, comm_library=CL, overlap=int(os.environ.get("COMM_OVERLAP", "1")))
    run.init_simulation()
    failed_at, msg = None, ""
    mode = os.environ.get("POISON_RUN_STEPS", "")
    if mode == "batch":
        # the same through rgpu_comm_run_steps: the failure sits in the first step of a call of three -- a host-driven step at step 0,
        # the first of a batch of three later on (the failed rank keeps pairing the collectives of the steps its peers have queued)
        if fstep:
            assert run.run_steps(fstep) == fstep
        if rank == frank:
            fail_next()
        try:
            run.run_steps(3)
        except Exception as e:  # noqa: BLE001
            failed_at, msg = run.nStep, str(e)
    elif mode == "single":
        # one call per step: from step 1 on every call is a batch of ONE step, the failure sits in its last step and the peers
        # leave their batch in good health -- the failed rank has to pair the all-reduce of THEIR next call
        for n in range(fstep + 3):
            if rank == frank and n == fstep:
                fail_next()
            try:
                run.run_steps(1)
            except Exception as e:  # noqa: BLE001
                failed_at, msg = run.nStep, str(e)
                break
    else:
        for n in range(fstep + 3):
            if rank == frank and n == fstep:
                fail_next()
            try:
                run.oneStepIntegration()
            except Exception as e:  # noqa: BLE001
                failed_at, msg = n, str(e)
                break
    flags = [None] * world
    dist.all_gather_object(flags, (failed_at, msg))
    # (the library's loop all-reduces the next step's 1/dt at the end of a step: a healthy rank sees the poison inside step fail_step)
    late = (fstep, fstep + 1) if os.environ.get("POISON_RUN_STEPS") else (fstep + 1,)
    good = all((f[0] == fstep) if r == frank else (f[0] in late) for r, f in enumerate(flags))
    good = good and all("not finite" in f[1] for r, f in enumerate(flags) if r != frank)
    if rank == 0:
        with open(out, "w") as f:
            f.write("OK\n" if good else "FAILED %r\n" % (flags,))
    dist.barrier()
    run.close()
    dist.destroy_process_group()
    sys.exit(0 if good else 1)


def main():
    if sys.argv[1] == "--frontend":
        return frontend()
    if sys.argv[1] == "--poison":
        return poison()
    if sys.argv[1] == "--batch":
        return batch()
    sys.exit(0 if one_case(sys.argv[1:5]) else 1)


def batch():
    """--batch spec.json: [{"argv": [base, overrides, nsteps, resultfile], "env": {...}}, ...] -- the cases of one test function that
    share a rank count, in ONE set of rank processes (interpreter start, torch import and the HIP context are most of a small
    case's wall time).  Every case writes its own result file; the exit code says whether all of them passed."""
    import json
    spec = json.load(open(sys.argv[2]))
    dist.init_process_group("gloo")
    good = True
    for n, case in enumerate(spec):
        saved = dict(os.environ)
        os.environ.update(case.get("env", {}))
        if dist.get_rank() == 0:
            print("### case %d: %s" % (n, " ".join(case["argv"][:3])), flush=True)
        try:
            good = one_case(case["argv"]) and good
        finally:
            os.environ.clear()
            os.environ.update(saved)
        dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if good else 1)


def one_case(argv):
    base, ov, nsteps, out = argv[0], argv[1], int(argv[2]), argv[3]
    alone = not dist.is_initialized()
    if alone:
        dist.init_process_group("gloo")
    try:
        return _one_case(base, ov, nsteps, out)
    finally:
        if alone:
            dist.destroy_process_group()


def _one_case(base, ov, nsteps, out):
    rank, world = dist.get_rank(), dist.get_world_size()
    device = os.environ.get("COMM_DEVICE", "cpu")
    lib, CL, keep = load_libs(device)
    ids = [rcomm.unique_id(CL) if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    ini = os.path.join(ROOT, "configs", base + ".ini")
    # world = 1 with periodic z: a ring of ONE rank -- the slab is its own z neighbour and its halo planes really go through the
    # transport (run.slabSelfRing); asserted below through the bytes the driver says it sends per exchange
    p0 = lib.params_from_ini(ini, ov)
    ring1 = world == 1 and p0.nz_global != 1 and p0.bc[4] == _capi.BC_PERIODIC and p0.bc[5] == _capi.BC_PERIODIC
    run = rcomm.CommRun(ini, ov, rank, world, ids[0], library=lib, comm_library=CL, overlap=int(os.environ.get("COMM_OVERLAP", "1")), self_ring=ring1)
    if ring1 or world > 1:
        gw = run.p.ghostWidth
        faces = int(run.p.bc[4] == _capi.BC_COPY) + int(run.p.bc[5] == _capi.BC_COPY)
        want = faces * gw * (run.p.nx + 2 * gw) * (run.p.ny + 2 * gw) * run.p.nbVar * 8
        assert run.halo_bytes() == want and (want > 0 or not ring1), (run.halo_bytes(), want)
    # the schedule in force: what the test asked for, or (COMM_OVERLAP=-1, the driver's choice) boundary-first for 3D MHD slabs of up to
    # 96 planes, overlapped otherwise (include/rgpu_comm.h; RGPU_COMM_SCHEDULE is not set by the tests that leave the choice open)
    asked = int(os.environ.get("COMM_OVERLAP", "1"))
    want_sched = asked if asked >= 0 else (int(os.environ["RGPU_COMM_SCHEDULE"]) if os.environ.get("RGPU_COMM_SCHEDULE") in ("1", "2") else (2 if (run.p.mhdEnabled and run.p.nz <= 96) else 1))
    assert run.schedule() == want_sched, (run.schedule(), want_sched)
    run.init_simulation()
    # COMM_RUN_STEPS=k: the steps through rgpu_comm_run_steps in two calls (k, then the rest) instead of one oneStepIntegration per step:
    # where the configuration allows it the time step stays on the device between steps and the host reads a batch of records once
    pieces = int(os.environ.get("COMM_RUN_STEPS", "0"))
    if pieces:
        dts = []
        for k in (min(pieces, nsteps), nsteps - min(pieces, nsteps)):
            if k:
                assert run.run_steps(k) == k, (k, run.nStep)
                dts += run.dt_log
        assert run.nStep == nsteps and len(dts) == nsteps and run.dt == dts[-1]
        # which loop ran: everything but the first step from the device record, unless the configuration keeps the host loop
        host_loop = (os.environ.get("COMM_OVERLAP", "1") == "0" or run.p.nu > 0 or (run.p.mhdEnabled and run.p.eta > 0) or run.p.randomForcingEnabled
                     or run.p.ouForcingEnabled or run.p.gravityEnabled != 0 or os.environ.get("RGPU_TEST_NO_STEP_CLOCK")
                     or (device == "cpu" and not run.p.mhdEnabled))   # (emulation: the 3D hydro pieces cannot carry the CFL scan without the tiled sweep)
        want_clocked = 0 if host_loop else nsteps - 1
        if os.environ.get("COMM_EXPECT_CLOCK", "1") == "1":
            assert run.clocked_steps() == want_clocked, (run.clocked_steps(), want_clocked)
        t_sum = 0.0
        for d in dts:
            t_sum += d
        assert run.totalTime == t_sum
    else:
        dts = [run.oneStepIntegration() for _ in range(nsteps)]
    staged = device.startswith("cuda-staged")
    if staged:
        # the wire really carried planes: one exchange per step (+ the one of the initial ghost fill), and with the packed exchange
        # ONE message per neighbour and exchange (two when the two neighbours are distinct ranks)
        st = (C.c_long * 3)()
        CL.rgpu_comm_test_stats(st)
        want_pack = os.environ.get("RGPU_COMM_PACK", "1") != "0"
        faces = int(run.p.bc[4] == _capi.BC_COPY) + int(run.p.bc[5] == _capi.BC_COPY)
        peers = 0 if faces == 0 else (1 if (world <= 2 or faces == 1) else 2)
        per_exchange = peers if want_pack else faces * run.p.nbVar
        assert st[2] == int(want_pack), list(st)
        assert st[0] >= nsteps + 1 and st[1] == st[0] * per_exchange, (list(st), per_exchange)
    # N-independent fingerprint (bench.py prints the same): dt sequence + the sum mod 2^64 of the slabs' state checksums
    sums = [None] * world
    dist.all_gather_object(sums, run.solver.state_checksum(run.nStep % 2))
    fingerprint = sum(sums) % (1 << 64)
    if staged and os.environ.get("COMM_CHECK", "oracle") == "single":
        # big boxes (the oracle would take minutes): against the single-device run of the whole box through the same library
        ok = True
        chunk = run.local_interior()
        if rank == 0:
            from ramsesgpu_amd.solver import Solver
            p = lib.params_from_ini(ini, ov)
            one = Solver(p, lib)
            one.upload(lib.init_condition(ini, ov, p), both=False)
            one.make_all_boundaries(0, 0.0, 0.0)
            dts_one = [one.oneStepIntegration() for _ in range(nsteps)]
            fp_one = one.state_checksum(one.nStep % 2)
            gw = p.ghostWidth
            ref0 = one.getDataHost(one.nStep % 2)[:, gw:-gw, gw:-gw, gw:-gw]
            one.close()
            nzl = p.nz // world
            nbad = int((chunk != ref0[:, :nzl]).sum())
            ok = nbad == 0 and dts == dts_one and fp_one == fingerprint
            msg = "rank 0: %d doubles differ, dt equal=%s, fingerprint %x / %x" % (nbad, dts == dts_one, fingerprint, fp_one)
            box = [ref0]
        else:
            box = [None]
        # the other ranks compare their own planes (the whole box does not travel through gloo three times)
        flags = [None] * world
        if rank == 0:
            for r in range(1, world):
                dist.send(torch.from_numpy(np.ascontiguousarray(box[0][:, r * nzl:(r + 1) * nzl])), dst=r)
            mine = (ok, msg)
        else:
            ref = torch.empty(chunk.shape, dtype=torch.float64)
            dist.recv(ref, src=0)
            nbad = int((chunk != ref.numpy()).sum())
            mine = (nbad == 0, "rank %d: %d doubles differ" % (rank, nbad))
        dist.all_gather_object(flags, mine)
        ok = all(f[0] for f in flags)
        if rank == 0:
            with open(out, "w") as f:
                f.write("OK %016x\n" % fingerprint if ok else "MISMATCH %r\n" % (flags,))
        dist.barrier()
        run.close()
        return ok
    local = torch.from_numpy(np.ascontiguousarray(run.local_interior()))
    parts = [torch.empty_like(local) for _ in range(world)] if rank == 0 else None
    dist.gather(local, parts, dst=0)
    ok = True
    if rank == 0:
        got = torch.cat(parts, dim=1).numpy()
        oracle = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
        p = lib.params_from_ini(ini, ov)
        U0 = lib.init_condition(ini, ov, p)
        oracle.set_gravity_field(lib.init_gravity(ini, ov, p))
        oracle.set_forcing_field(lib.init_forcing(ini, ov, p))
        ref_full, dts_ref, _ = oracle.run(p, U0, nsteps)
        ref = interior(ref_full, p)
        nbad = int((got != ref).sum())
        ok = nbad == 0 and np.array_equal(np.array(dts), dts_ref)
        if p.randomForcingEnabled or (p.ouForcingEnabled and device != "cpu"):   # (OU on a GPU: the device's cos())   # global normalisation sum: round-off agreement (stated tolerance 1e-12), see slab_worker.py
            rel = float(np.sqrt(((got - ref) ** 2).sum() / (ref ** 2).sum()))
            ok = rel < 1e-12 and np.allclose(np.array(dts), dts_ref, rtol=1e-12, atol=0)
        box = [dts_ref.tolist() if pieces else None]
        if ok and not (p.randomForcingEnabled or p.ouForcingEnabled):   # the fingerprint is the oracle's own
            ok = fingerprint == int(np.ascontiguousarray(ref).view(np.uint64).sum(dtype=np.uint64))
        with open(out, "w") as f:
            f.write("OK %016x\n" % fingerprint if ok else "MISMATCH %d doubles, dt equal=%s, fingerprint %x\n" % (nbad, np.array_equal(np.array(dts), dts_ref), fingerprint))
    if pieces and nsteps >= 2:
        # an end time inside a batch: the loop condition "t < tEnd" is evaluated on the device, the steps queued behind it are no-ops on
        # every rank; afterwards the state, its ghost planes and its CFL maxima are those of the last step that ran
        box = box if rank == 0 else [None]
        dist.broadcast_object_list(box, src=0)
        dts_ref_l = box[0]
        cut = max(1, nsteps // 2)
        t_cut = 0.0
        for d in dts_ref_l[:cut]:
            t_cut += d
        tEnd = t_cut - 0.25 * dts_ref_l[cut - 1]
        run.close()
        ids = [rcomm.unique_id(CL) if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        run = rcomm.CommRun(ini, ov, rank, world, ids[0], library=lib, comm_library=CL, overlap=int(os.environ.get("COMM_OVERLAP", "1")), self_ring=ring1)
        run.init_simulation()
        done = run.run_steps(nsteps + 3, tEnd)
        good = done == cut and run.nStep == cut and run.totalTime == t_cut and run.dt == dts_ref_l[cut - 1]
        local = torch.from_numpy(np.ascontiguousarray(run.local_interior()))
        parts = [torch.empty_like(local) for _ in range(world)] if rank == 0 else None
        dist.gather(local, parts, dst=0)
        good = good and run.run_steps(5, tEnd) == 0
        dt_next = run.oneStepIntegration()                      # the state is usable: same next dt as the uninterrupted run
        good = good and (cut >= len(dts_ref_l) or dt_next == dts_ref_l[cut])
        flags = [None] * world
        dist.all_gather_object(flags, (good, done, run.totalTime))
        if rank == 0:
            ref_cut, _, _ = oracle.run(p, U0, cut)
            nbad2 = int((torch.cat(parts, dim=1).numpy() != interior(ref_cut, p)).sum())
            ok2 = all(f[0] for f in flags) and nbad2 == 0
            if not ok2:
                with open(out, "w") as f:
                    f.write("MISMATCH end time inside the batch: %d doubles differ, per rank (ok, steps, t) = %r, expected %d steps, t = %r\n" % (nbad2, flags, cut, t_cut))
            ok = ok and ok2
    oks = [None] * world
    dist.all_gather_object(oks, bool(ok))
    run.close()
    return all(oks)


if __name__ == "__main__":
    main()
