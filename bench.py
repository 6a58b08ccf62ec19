#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json: Mcell-updates/s of the 3D MHD unsplit step (fp64).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is one oneStepIntegration of the hot path = compute_dt_mhd + godunov_unsplit (rotating + shearing-box
variant) incl. the ghost fill, on synthetic data: configs/mhd_mri_3d.ini (values of the reference's
data/mhd_mri_3d.ini) scaled to 512^3, [MRI] seed=0 -- BASELINE.json configs[3].  The state is resident in HBM
when the timed region starts.  With N>1 the SAME 512^3 box is cut into N z-slabs (strong scaling), one process per
GPU, halo planes exchanged with RCCL point-to-point calls, 1/dt max-reduced with one all-reduce.

value = K * nx*ny*nz / t / 1e6 with t = max over ranks of the wall time of the K steps, bracketed by
barrier + torch.cuda.synchronize() on both sides: the reference's own "cell updates per second"
(MHDRunGodunov.cpp:4064-4068).

Extra objects on the JSON line:
  roofline      dominant kernel of the step (by accumulated time): algorithmic bytes per launch (128 B per cell
                update: read U once + write U once) / its average duration measured with
                HIP events on the kernel's stream, against the 8 TB/s HBM3E peak.  `traffic` comes from the
                rocprofv3 --pmc summary under profiles/ (null if absent).
  roofline_step the same accounting for the whole step (all kernels).
  cpu_baseline  N=1, rank 0 only: the reference binary oracle/_ref/euler_cpu ("reference") -- or the oracle's
                restatement ("port") -- on ONE host core, on a bounded sample of the same workload.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BASE = "mhd_mri_3d"
HBM_PEAK = 8.0e12                 # B/s, MI355X HBM3E (guide: MI355X_MICROARCH.md)
ALGO_BYTES_PER_CELL = 128.0       # MHD: 8 stored variables, read once + written once, fp64


def workload_overrides(n):
    return "mesh.nx=%d;mesh.ny=%d;mesh.nz=%d" % (n, n, n)


def _ref_ini(size, steps):
    ini_text = open(os.path.join(ROOT, "configs", BASE + ".ini")).read()
    for k, v in (("nx", size), ("ny", size), ("nz", size), ("nstepmax", steps), ("noutput", 10 ** 6),
                 ("outputVtk", "no"), ("outputHdf5", "no")):
        ini_text = re.sub(r"(?m)^%s=.*$" % k, "%s=%s" % (k, v), ini_text)
    return ini_text


def _ref_rates(ref_bin, size, steps, copies):
    """run `copies` independent euler_cpu processes at once; returns their reported cell-update rates [1/s] and the wall time"""
    with tempfile.TemporaryDirectory() as td:
        procs = []
        t0 = time.time()
        for c in range(copies):
            d = os.path.join(td, "r%d" % c)
            os.makedirs(d)
            open(os.path.join(d, "b.ini"), "w").write(_ref_ini(size, steps))
            procs.append(subprocess.Popen([ref_bin, "--param", "b.ini"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                          universal_newlines=True))
        outs = [pr.communicate()[0] for pr in procs]
        wall = time.time() - t0
    rates = []
    for out in outs:
        m = re.search(r"([0-9.eE+-]+) cell updates per seconds", out)
        if m:
            rates.append(float(m.group(1)))
    return rates, wall


def cpu_baseline_all_cores(size, steps):
    """the reference binary is single-threaded (its OpenMP build races): occupy the host with one
    independent replica per core and add the rates up -- an upper bound for any domain-decomposed CPU run"""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "euler_cpu")
    if not os.path.exists(ref_bin):
        return None
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        avail_gb = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE") / 2.0 ** 30
    except (ValueError, OSError):
        avail_gb = 16.0
    per_copy_gb = 1.7e-6 * (size + 6) ** 3      # ~1.7 kB per cell in the reference's 3D MHD arrays
    copies = int(max(1, min(cores, 64, 0.5 * avail_gb / per_copy_gb)))   # 64 replicas already saturate the host's memory system
    rates, wall = _ref_rates(ref_bin, size, steps, copies)
    if len(rates) != copies:
        return None
    return {"value": sum(rates) / 1e6, "unit": "Mcell-updates/s", "cores": copies, "kind": "reference",
            "sample": "%d independent replicas of %s at %d^3, %d steps, one single-threaded euler_cpu per core, rates summed (%.1f s)"
                      % (copies, BASE, size, steps, wall)}


def cpu_baseline(size, steps):
    """time the CPU path on a bounded sample (same physics, smaller box; the metric is intensive)"""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "euler_cpu")
    sample = "%s at %d^3, %d steps, 1 thread" % (BASE, size, steps)
    if os.path.exists(ref_bin):
        rates, wall = _ref_rates(ref_bin, size, steps, 1)
        if rates:
            return {"value": rates[0] / 1e6, "unit": "Mcell-updates/s", "cores": 1, "kind": "reference",
                    "sample": sample + " (oracle/_ref/euler_cpu, g++ -O2, %.1f s)" % wall}
    # fall back to the oracle's restatement (bit-identical arithmetic, same loop structure)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_api import Oracle
    from ramsesgpu_amd.solver import load_library
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    L = load_library()
    ini = os.path.join(ROOT, "configs", BASE + ".ini")
    p = L.params_from_ini(ini, workload_overrides(size))
    U0 = L.init_condition(ini, workload_overrides(size), p)
    t0 = time.time()
    Oracle(so).run(p, U0, steps)
    wall = time.time() - t0
    return {"value": steps * size ** 3 / wall / 1e6, "unit": "Mcell-updates/s", "cores": 1, "kind": "port",
            "sample": sample + " (oracle/liboracle.so, g++ -O2, %.1f s)" % wall}


def pmc_traffic(kernel_phase):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc summary, or None"""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    try:
        d = json.load(open(path))
        return d.get(kernel_phase, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=512, help="box edge (default: the 512^3 headline workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timeline-only", action="store_true", help="stop after the timed region (for rocprofv3 --kernel-trace concurrency analysis)")
    ap.add_argument("--cpu-size", type=int, default=96)
    ap.add_argument("--cpu-steps", type=int, default=10)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from ramsesgpu_amd.slab import SlabRun
    from ramsesgpu_amd.solver import Solver, load_library

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        if rank == 0:
            sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)\n" % (args.gpus, world))
        sys.exit(2)
    if not torch.cuda.is_available():
        sys.stderr.write("bench.py: no GPU visible; the product has no CPU fallback\n")
        sys.exit(3)
    # test hooks (not used by the driver): RGPU_BENCH_ONE_DEVICE=1 puts every rank on GPU 0 and RGPU_BENCH_BACKEND=gloo
    # replaces RCCL, which refuses two ranks on one device -- lets the N>1 code path be exercised on a 1-GPU box
    if os.environ.get("RGPU_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("RGPU_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    L = load_library()
    ini = os.path.join(ROOT, "configs", BASE + ".ini")
    ov = workload_overrides(args.size)
    n = args.size

    if world == 1:
        p = L.params_from_ini(ini, ov)
        U0 = L.init_condition(ini, ov, p)
        run = Solver(p, L)
        run.upload(U0, both=False)
        del U0
        run.make_all_boundaries(0, 0.0, 0.0)
        # (the reference's h_U.copyTo(h_U2) is not needed: every step writes the whole output array)
        step = run.oneStepIntegration
        timers_src = run
    else:
        srun = SlabRun(ini, ov, library=L, device="cuda:%d" % local_rank)
        srun.init_simulation()
        step = srun.oneStepIntegration
        timers_src = srun.solver
        p = srun.p

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    timers_src.enable_timers(False)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if args.timeline_only:
        if rank == 0:
            print(json.dumps({"ms_per_step": elapsed / args.steps * 1e3}))
        return
    # per-kernel durations: the same steps again with HIP events around every launch on the kernels' stream
    # (separate from the timed region: the events serialise host and device)
    timers_src.enable_timers(True)
    timers_src.reset_timers()
    nprof = min(5, max(args.steps, 1))
    for _ in range(nprof):
        step()
    sync()
    tm = timers_src.timers()
    dom_name, dom_ms, dom_launches = timers_src.dominant_kernel()
    timers_src.enable_timers(False)

    cells_global = float(n) ** 3
    cells_local = cells_global / world
    value = args.steps * cells_global / elapsed / 1e6
    if rank == 0:
        step_bytes = ALGO_BYTES_PER_CELL * cells_local
        # per step: a slab run launches the kernel once per plane range (two boundary ranges + the inner one)
        dom_ms = dom_ms * dom_launches / nprof
        achieved = step_bytes / (dom_ms * 1e-3)
        step_ms_events = sum(tm.values()) / nprof * 1e3
        out = {
            "metric": "Mcell-updates/s", "value": value, "unit": "Mcell-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs/mhd_mri_3d.ini (= reference data/mhd_mri_3d.ini) scaled to %d^3: 3D MHD MRI, "
                                   "isothermal, rotating frame + shearing box, HLLD + MAG_HLLD + CT, [MRI] seed=0" % n,
                       "nx": n, "ny": n, "nz": n, "decomposition": "z-slabs x%d" % world,
                       "path": "compute_dt_mhd + godunov_unsplit (rotating) + shearing ghost fill",
                       "parity": "bit-identical to euler_cpu on all golden fixtures (tests/)"},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": pmc_traffic(dom_name),
                         "algorithmic_bytes_per_launch": step_bytes, "avg_launch_ms": dom_ms, "launches_per_step": dom_launches / nprof, "launches_timed": dom_launches,
                         "note": "ALU-bound kernel (fp64 div/sqrt heavy Riemann solvers): see DESIGN.md"},
            "roofline_step": {"bound": "hbm", "achieved": step_bytes / (elapsed / args.steps) / 1e9, "peak": HBM_PEAK / 1e9,
                              "unit": "GB/s", "frac": step_bytes / (elapsed / args.steps) / HBM_PEAK,
                              "phase_ms": {k: v / nprof * 1e3 for k, v in tm.items() if v > 0}, "sum_phase_ms": step_ms_events},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_size, args.cpu_steps)
            allc = cpu_baseline_all_cores(64, 10)
            if allc:
                out["cpu_baseline_all_cores"] = allc
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
