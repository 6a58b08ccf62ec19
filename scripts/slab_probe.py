#!/usr/bin/env python3
"""Per-rank cost of the z-slab schedule on ONE GPU: one rank of a ring of one ([run] slabSelfRing: the rank is its own z neighbour,
the halo planes really travel through RCCL send / recv on the halo stream, device-local instead of over xGMI; until round 4 a
ring of one had NO slab interface and this probe exchanged nothing -- rgpu_comm_halo_bytes now proves the bytes) stepping a
512 x 512 x (512/N) box through the C++ driver (include/rgpu_comm.h) with the overlapped and the serial schedule -- what a
rank of bench.py --gpus N does per step.  The device-local self-exchange costs ~0.5 ms for the 103 MB; with a link rate the probe loads the
MEASUREMENT build of the driver (ramsesgpu_amd/librgpu_comm_measure.so: `python -m ramsesgpu_amd.build --measure`, built in the container
before the gpurun call; scripts/measure/link_hold.h) whose RGPU_COMM_EMULATE_GBPS=<rate> holds the halo stream for the time the same
bytes need on ONE xGMI link at that rate
(2 x 51.5 MB per rank and step at 512^2 planes: N >= 3 -> two neighbours, two links in parallel, 51.5 MB each; N = 2 -> one
neighbour, 103 MB over one link), so that the numbers show what the overlapped schedule really hides.  PROBE_LINK_GBPS="0 60 40"
runs the whole table once per rate."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ramsesgpu_amd import comm as rcomm
from ramsesgpu_amd.solver import load_library
L = load_library()
rates = [float(x) for x in os.environ.get("PROBE_LINK_GBPS", os.environ.get("RGPU_COMM_EMULATE_GBPS", "0")).split()]
measure = os.path.join(ROOT, "ramsesgpu_amd", "librgpu_comm_measure.so")
if any(r > 0 for r in rates):
    assert os.path.exists(measure) and L.arithmetic == "contracted", "link rates need the measurement build (python -m ramsesgpu_amd.build --measure) and RGPU_ARITH=contracted"
    CL = rcomm.load_comm_library(measure)
else:
    CL = rcomm.load_comm_library(rcomm.comm_lib_path(L.arithmetic))   # the driver built for this arithmetic (RGPU_ARITH)
ini = os.path.join(ROOT, "configs", "mhd_mri_3d.ini")
cid = rcomm.unique_id(CL)
for rate, nz in [(r, z) for r in rates for z in ([int(os.environ['PROBE_NZ'])] if os.environ.get('PROBE_NZ') else (512, 256, 128, 64))]:   # PROBE_NZ: one slab thickness (for rocprofv3)
    if rate > 0 and nz == 512:
        continue   # N = 1 has no link
    os.environ.pop("RGPU_COMM_EMULATE_GBPS", None)
    if rate > 0:
        os.environ["RGPU_COMM_EMULATE_GBPS"] = "%g" % rate
        os.environ["RGPU_COMM_EMULATE_PEERS"] = "1" if 512 // nz == 2 else "2"
    for overlap in ((None,) if os.environ.get('PROBE_NZ') else (2, 1, 0)):   # schedule of include/rgpu_comm.h (None: the driver's choice)
        run = rcomm.CommRun(ini, "mesh.nx=512;mesh.ny=512;mesh.nz=%d" % nz, 0, 1, cid, library=L, comm_library=CL, overlap=overlap, self_ring=True)
        assert run.halo_bytes() == 2 * 3 * 518 * 518 * 8 * 8, run.halo_bytes()   # the planes really go through RCCL
        run.init_simulation()
        for _ in range(3): run.oneStepIntegration()
        run.solver.synchronize(); t0 = time.time()
        n = 10
        if os.environ.get("PROBE_HOST_LOOP"):      # rounds 1-4: one host turn per step
            for _ in range(n): run.oneStepIntegration()
        else:                                      # round 5: rgpu_comm_run_steps, the time step on the device between the steps
            k = int(os.environ.get("PROBE_BATCH", n))   # steps per rgpu_comm_run_steps call
            for _ in range(n // k):
                assert run.run_steps(k) == k
            assert run.clocked_steps() == (0 if overlap == 0 else (n // k) * k), run.clocked_steps()   # (the serial schedule keeps the host loop)
        run.solver.synchronize(); dt = (time.time() - t0) / n
        link_ms = 0.0 if rate <= 0 else 8 * 518 * 518 * 3 * 8 * (2 if 512 // nz == 2 else 1) / rate / 1e6
        mode = "%s" % ("link time beside the local copy" if os.environ.get("RGPU_COMM_EMULATE_MODE") == "parallel" else "link time behind the local copy") if rate > 0 else ""
        print("nz=%3d (N=%d) link %3g GB/s (%.2f ms per exchange) %-8s %7.2f ms/step  -> %6.0f Mcell/s per rank, x%d = %6.0f" % (nz, 512 // nz, rate, link_ms,
              {None: "default", 2: "bnd-first", 1: "overlap", 0: "serial"}[overlap], dt * 1e3, 512 * 512 * nz / dt / 1e6, 512 // nz, 512 * 512 * 512 / dt / 1e6) + ("  [%s]" % mode if mode else ""), flush=True)
        run.close()
        cid = rcomm.unique_id(CL)
