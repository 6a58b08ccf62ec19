#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json: Mcell-updates/s of the unsplit Godunov step (fp64).

    python bench.py --gpus N --steps K --warmup W [--workload mri|implode3d|orszag-tang] [--size S | --nx --ny --nz]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is one oneStepIntegration of the hot path = compute_dt[_mhd] + godunov_unsplit incl. the ghost fill, on synthetic
data; the state is resident in HBM when the timed region starts.  Workloads (BASELINE.json configs):
  mri          (default; the headline line) configs/mhd_mri_3d.ini scaled to 512^3: 3D MHD shearing box, isothermal,
               HLLD + MAG_HLLD + CT, [MRI] seed=0 -- configs[3]; --nx 512 --ny 1024 --nz 512 --gpus 8 is configs[4]
  implode3d    configs/implode3d.ini at 256^3 with riemannSolver=hllc -- configs[1]
  orszag-tang  configs/orszag-tang.ini as shipped (512^2, 2D MHD) -- configs[2]; its correctness gate is in tests/
With N>1 the SAME 3D box is cut into N z-slabs (strong scaling), one process per GPU, halo planes exchanged with RCCL
point-to-point calls, 1/dt max-reduced with one all-reduce.  2D boxes do not shard: N independent replicas (weak).

value = K * nx*ny*nz / t / 1e6 with t = max over ranks of the wall time of the K steps, bracketed by
barrier + torch.cuda.synchronize() on both sides: the reference's own "cell updates per second"
(MHDRunGodunov.cpp:4064-4068).

Extra objects on the JSON line:
  roofline      dominant kernel of the step (by accumulated time): algorithmic bytes per launch (read U once + write U
                once: 128 B per MHD cell update, 80 B hydro 3D) / its average duration measured
                with HIP events on the kernel's stream, against the 8 TB/s HBM3E peak.  `traffic` comes from the
                rocprofv3 --pmc summary under profiles/ (null if absent): FETCH_SIZE + WRITE_SIZE as counted, a lower
                bound on gfx950; `traffic_fetch_doubled` = with the guide's x2 on FETCH_SIZE, an upper bound.
  roofline_step the same accounting for the whole step (all kernels).
  cpu_baseline  N=1, rank 0 only: the reference binary oracle/_ref/euler_cpu ("reference") -- or the oracle's
                restatement ("port") -- on ONE host core, on a bounded sample of the same workload (256^3 for the 3D
                workloads when the host has the memory).
  cpu_baseline_all_cores  ONE 256^3 box on all host cores: the restatement threaded over z-slabs (deterministic gather
                update); `upper_bound_replicas` = independent 64^3 replicas of the reference binary, rates summed.
  config.arithmetic  which build of the library `value` was measured with.  Default "contracted": librgpu_fast.so, the
                tolerance-grade build (FMA contraction, ~1-ulp division / square root), held to north_star's bar -- relative L2
                < 1e-12 to euler_cpu -- by tests/test_contracted.py (all fixtures, the full 512^2 x 50 Orszag-Tang gate, 300-400 step
                runs, the headline-size property checks; worst measured 2e-14).  --arith exact: librgpu.so, bit-identical.
  value_exact   (every N, default arithmetic) the same workload, same K and W, through the bit-identical library librgpu.so (N=1: with its
                own roofline; N>1: a second RCCL communicator after the first is destroyed).  --arith exact prints the contracted record
                beside it as value_tolerance instead.  value_arithmetic / metric_version say which build `value` is.
  other_workloads  default headline run only: short measurements of BASELINE configs[1] (implode3d 256^3) and [2]
                (orszag-tang 512^2) with their own roofline / cpu_baseline.
  config.fingerprint  every N: sha256 of the dt sequence of all W + K steps + the sum mod 2^64 of the bit patterns of every interior double of
                the final state (rgpu_state_checksum per rank, added up) -- identical for every --gpus N at the same arithmetic, box, K and W,
                because the slab driver reproduces the single-device run bit for bit: a SCALE line can be checked against the N = 1 line.
                value_exact / value_tolerance carry their own.
  config.driver / rccl_ranks / ranks  which slab driver ran, what RCCL itself reports (ncclCommCount), device + PCI bus id per
                rank.  --gpus N > 1 has NO fallback: it is the C++ RCCL driver or a non-zero exit.
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
# before anything can initialise the HSA runtime: the host driver of this pool only supports dmabuf IPC (RCCL's peer buffers)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK = 8.0e12                 # B/s, MI355X HBM3E (guide: MI355X_MICROARCH.md)

WORKLOADS = {
    # name: ini base, extra overrides, default (nx, ny, nz), algorithmic bytes per cell update, description, path
    "mri": dict(base="mhd_mri_3d", extra="", size=(512, 512, 512), bytes=128.0,
                desc="configs/mhd_mri_3d.ini (= reference data/mhd_mri_3d.ini) scaled to %s: 3D MHD MRI, isothermal, "
                     "rotating frame + shearing box, HLLD + MAG_HLLD + CT, [MRI] seed=0",
                path="compute_dt_mhd + godunov_unsplit (rotating) + shearing ghost fill",
                ref_mb_per_cell=1.7e-3),
    "implode3d": dict(base="implode3d", extra="hydro.riemannSolver=hllc", size=(256, 256, 256), bytes=80.0,
                      desc="configs/implode3d.ini (= reference data/implode3d.ini) at %s: 3D hydro implosion, HLLC, "
                           "reflecting walls",
                      path="compute_dt + godunov_unsplit (hydro unsplit version 1) + ghost fill",
                      ref_mb_per_cell=0.6e-3),
    "orszag-tang": dict(base="orszag-tang", extra="", size=(512, 512, 1), bytes=128.0,
                        desc="configs/orszag-tang.ini (= reference data/orszag-tang.ini) at %s: 2D MHD Orszag-Tang vortex, "
                             "HLLD + MAG_HLLD + CT",
                        path="compute_dt_mhd + godunov_unsplit (2D MHD implementation 1) + ghost fill",
                        ref_mb_per_cell=0.8e-3),
}


def overrides_for(w, nx, ny, nz):
    ov = "mesh.nx=%d;mesh.ny=%d;mesh.nz=%d" % (nx, ny, nz)
    return ov + (";" + w["extra"] if w["extra"] else "")


def _ref_ini(w, nx, ny, nz, steps):
    ini_text = open(os.path.join(ROOT, "configs", w["base"] + ".ini")).read()
    edits = [("nx", nx), ("ny", ny), ("nz", nz), ("nstepmax", steps), ("noutput", 10 ** 6), ("outputVtk", "no"), ("outputHdf5", "no")]
    for kv in (w["extra"].split(";") if w["extra"] else []):
        k, v = kv.split("=")
        edits.append((k.split(".")[1], v))
    for k, v in edits:
        ini_text = re.sub(r"(?m)^%s=.*$" % k, "%s=%s" % (k, v), ini_text)
    return ini_text


def _ref_rates(ref_bin, w, dims, steps, copies):
    """run `copies` independent euler_cpu processes at once; returns their reported cell-update rates [1/s] and the wall time"""
    with tempfile.TemporaryDirectory() as td:
        procs = []
        t0 = time.time()
        for c in range(copies):
            d = os.path.join(td, "r%d" % c)
            os.makedirs(d)
            open(os.path.join(d, "b.ini"), "w").write(_ref_ini(w, dims[0], dims[1], dims[2], steps))
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


def _avail_gb():
    try:
        return os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE") / 2.0 ** 30
    except (ValueError, OSError):
        return 16.0


def cpu_sample(w, dims, budget_s=25.0):
    """bounded sample of the workload for the 1-core CPU leg: 256^3 for the 3D boxes when the host has the memory for the
    reference's arrays, else 128^3; the 2D workload as shipped.  Steps sized for ~budget_s at the
    reference's ~0.7 (MHD) / ~1.7 (hydro) Mcell-updates/s."""
    if dims[2] == 1:
        n = (min(dims[0], 512), min(dims[1], 512), 1)
        rate = 0.9e6
    else:
        e = 256 if _avail_gb() > 1.5 * w["ref_mb_per_cell"] * 262 ** 3 / 1024.0 else 128
        e = min(e, max(dims))
        n = (min(e, dims[0]), min(e, dims[1]), min(e, dims[2]))
        rate = 0.7e6 if w["bytes"] == 128.0 else 1.7e6
    cells = n[0] * n[1] * n[2]
    steps = int(max(3, min(50, round(budget_s * rate / cells))))   # at least three steps: one step is a noisy denominator
    if cells * 3 / rate > 1.6 * budget_s and dims[2] != 1:         # ... and a box they fit the budget with (3D: 192^3 instead of 256^3)
        e = 192 if n[0] > 192 else n[0]
        n = (min(e, n[0]), min(e, n[1]), min(e, n[2]))
    return n, steps


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(w, dims, budget_s=25.0):
    """time the CPU path on a bounded sample (same physics, smaller box; the metric is intensive)"""
    n, steps = cpu_sample(w, dims, budget_s)
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "euler_cpu")
    sample = "%s at %dx%dx%d, %d steps, 1 thread of %s" % (w["base"], n[0], n[1], n[2], steps, _cpu_model())
    if os.path.exists(ref_bin):
        rates, wall = _ref_rates(ref_bin, w, n, steps, 1)
        if rates:
            return {"value": rates[0] / 1e6, "unit": "Mcell-updates/s", "cores": 1, "kind": "reference",
                    "sample": sample + " (oracle/_ref/euler_cpu, g++ -O2, %.1f s incl. initial condition)" % wall}
    # fall back to the oracle's restatement (bit-identical arithmetic, same loop structure)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_api import Oracle
    from ramsesgpu_amd.solver import load_library
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    L = load_library()
    ini = os.path.join(ROOT, "configs", w["base"] + ".ini")
    ov = overrides_for(w, *n)
    p = L.params_from_ini(ini, ov)
    U0 = L.init_condition(ini, ov, p)
    t0 = time.time()
    Oracle(so).run(p, U0, steps)
    wall = time.time() - t0
    return {"value": steps * n[0] * n[1] * n[2] / wall / 1e6, "unit": "Mcell-updates/s", "cores": 1, "kind": "port",
            "sample": sample + " (oracle/liboracle.so, g++ -O2, %.1f s)" % wall}


def cpu_baseline_replicas(w, steps=10):
    """UPPER BOUND for any domain-decomposed CPU run of the reference: the reference binary is single-threaded (its OpenMP
    build races), so occupy the host with one independent 64^3 replica per core and add the rates up.  NOT a
    decomposed run of one box (no halo traffic, every replica cache-resident)."""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "euler_cpu")
    if not os.path.exists(ref_bin) or w["base"] != "mhd_mri_3d":
        return None
    size = 64
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    per_copy_gb = 1.7e-6 * (size + 6) ** 3      # ~1.7 kB per cell in the reference's 3D MHD arrays
    copies = int(max(1, min(cores, 64, 0.5 * _avail_gb() / per_copy_gb)))   # 64 replicas already saturate the host's memory system
    rates, wall = _ref_rates(ref_bin, w, (size, size, size), steps, copies)
    if len(rates) != copies:
        return None
    return {"value": sum(rates) / 1e6, "unit": "Mcell-updates/s", "cores": copies, "kind": "reference", "label": "upper_bound",
            "sample": "%d INDEPENDENT replicas of %s at %d^3, %d steps, one single-threaded euler_cpu per core, rates summed (%.1f s)"
                      % (copies, w["base"], size, steps, wall)}


def cpu_baseline_all_cores(w, dims):
    """ONE box of the workload on all host cores -- the oracle's restatement threaded over z-slabs (orc_run_mt:
    every loop nest of the 3D MHD step cut into contiguous slabs of planes, one std::thread each; the flux loop stores its
    fluxes and each cell gathers them in the order the reference's scatter loop delivers them, so the result is deterministic
    and bit-identical to the 1-thread run -- the reference's own OpenMP loops, mhd_godunov_unsplit_cpu_v3.cpp:32-35, 368-371,
    race on that scatter).  g++ -O2, no -march.  Threads are PINNED and every z-slab of the state and work arrays is first
    touched by the thread that works on it (orc_set_thread_placement): placement "all" = every allowed CPU in NUMA-node order,
    "node0" = the first NUMA node alone (one socket); for each a thread-count scan, the best of all is `value`, the rest is in
    `placements`.  3D MHD workloads only."""
    if w["bytes"] != 128.0 or dims[2] == 1:
        return None
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_api import Oracle
    from ramsesgpu_amd.solver import load_library
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    e = 256 if _avail_gb() > 1.5 * 1.7e-6 * 262 ** 3 else 128      # ~1.7 kB per cell: 167 work doubles + four state arrays
    n = (min(e, dims[0]), min(e, dims[1]), min(e, dims[2]))
    L = load_library()
    ini = os.path.join(ROOT, "configs", w["base"] + ".ini")
    ov = overrides_for(w, *n)
    p = L.params_from_ini(ini, ov)
    U0 = L.init_condition(ini, ov, p)
    O = Oracle(so)
    cells = n[0] * n[1] * n[2]
    more = 4
    runs = []
    for label, mode in (("all", 1), ("node0", 2)):
        ncpu = O.set_thread_placement(mode)
        if ncpu <= 0 or (label == "node0" and ncpu >= cores):   # no NUMA information, or a single node: "all" already is it
            continue
        # thread counts: the CPUs of the set down to 16 in halves, one step each; then `more` steps with the fastest
        scan, t = [], ncpu
        while t >= 16 or not scan:
            scan.append(t)
            t //= 2
        scan = [scan[0]] + scan      # (the first step of a run is cold: page tables, caches -- let it not decide)
        secs, used = O.run_mt_scan(p, U0, len(scan) + more, scan, ncpu)
        per_step = float(sum(secs[len(scan):]) / max(len(secs) - len(scan), 1))
        runs.append({"placement": label, "cpus_in_set": ncpu, "threads": used, "s_per_step": per_step, "value": cells / per_step / 1e6,
                     "scan": ", ".join("%d -> %.2f s" % (a, b) for a, b in zip(scan, secs))})
    O.set_thread_placement(0)
    if not runs:
        return None
    best = max(runs, key=lambda r: r["value"])
    out = {"value": best["value"], "unit": "Mcell-updates/s", "cores": best["threads"], "kind": "port",
           "sample": "ONE %s box at %dx%dx%d on %d pinned threads (placement %s: %d CPUs; host has %d), z-slabs of planes first touched by their "
                     "threads; oracle/liboracle.so orc_run_mt_scan, g++ -O2: %.2f s per step over %d steps; thread-count scan, one step each: %s"
                     % (w["base"], n[0], n[1], n[2], best["threads"], best["placement"], best["cpus_in_set"], cores, best["s_per_step"], more, best["scan"]),
           "placements": runs}
    ub = cpu_baseline_replicas(w)
    if ub:
        out["upper_bound_replicas"] = ub
    return out


_PMC = {}


def pmc_summary():
    """profiles/pmc_traffic.json, or None when it is absent OR was collected on another state of the kernel sources: the summary
    carries the hash of the device sources it was measured on (ramsesgpu_amd/build.py: kernel_source_hash, recorded by
    scripts/prof_round.sh); a kernel change without a PMC refresh must not silently skew `traffic` / `valu_ceiling`."""
    if "d" not in _PMC:
        _PMC["d"], _PMC["note"] = None, None
        path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        try:
            from ramsesgpu_amd import build as rb
            d = json.load(open(path))
            have, want = d.get("_kernel_source_sha"), rb.kernel_source_hash()
            if have == want:
                _PMC["d"] = d
            else:
                _PMC["note"] = ("profiles/pmc_traffic.json was collected on kernel sources %s, this build is %s: traffic / valu_ceiling withheld "
                                "(re-run scripts/prof_round.sh + scripts/summarize_prof.py)" % (have, want))
        except Exception as e:  # noqa: BLE001
            _PMC["note"] = "no usable profiles/pmc_traffic.json (%r)" % (e,)
    return _PMC["d"]


def pmc_traffic(workload, kernel_phase, doubled_fetch=False, scale=1.0):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc summary (separate FETCH_SIZE / WRITE_SIZE passes,
    KiB x 1024), or None.  On gfx950 FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section: exactly 1/2 for
    16 B per lane streaming reads, "other widths uncalibrated"); calibrated on this code's own 8 B per lane SoA streams against known byte
    counts it reports 0.49-0.77 of them.  Hence two figures: FETCH + WRITE as counted (a lower bound) and 2 x FETCH + WRITE (the guide's
    correction, an upper bound for these kernels)."""
    d = pmc_summary()
    if d is None:
        return None
    try:
        d = d.get(workload, d).get(kernel_phase, {})
        if doubled_fetch:
            return scale * (2.0 * d["fetch_bytes"] + d["write_bytes"])
        return scale * d["hbm_bytes_per_launch"]
    except Exception:
        return None


PARITY = {"exact": "librgpu.so: bit-identical to euler_cpu on all golden fixtures, oracle runs and the full Orszag-Tang gate (tests/)",
          "contracted": "librgpu_fast.so (FMA contraction, ~1-ulp division / sqrt; not bit-identical): relative L2 < 1e-12 to euler_cpu -- "
                        "north_star's tolerance -- gated by tests/test_contracted.py on all golden fixtures (worst 2e-14), the full 512^2 x 50 "
                        "Orszag-Tang gate (<= 9e-16 per variable), 300-400 step runs (<= 2.1e-15) and the headline-size property checks"}


def valu_ceiling(workload, kernel_phase, launch_ms, scale=1.0):
    """The second ceiling of SURVEY.md 8(d): share of the fp64 vector-issue capacity the dominant kernel uses.  Wave-level
    VALU instruction counts per launch come from the committed rocprofv3 --pmc summary (SQ_INSTS_VALU,
    SQ_INSTS_VALU_TRANS_F64); an fp64 instruction occupies a SIMD for 4 cycles (16 lanes per cycle), rcp / rsq / sqrt for 16;
    1024 SIMDs at the 2.4 GHz peak clock.  None without the summary.  scale: cells per launch of this run / of the profiled one."""
    try:
        d = pmc_summary()[workload][kernel_phase]
        insts, trans = scale * d["valu_wave_insts"], scale * d.get("valu_trans_f64_wave_insts", 0.0)
    except Exception:
        return None
    cycles = 4.0 * insts + 12.0 * trans
    cap = 1024 * 2.4e9 * launch_ms * 1e-3
    # what instruction streams reach on this chip at the kernel's occupancy (2 waves per SIMD), cycles per wave instruction at the
    # nominal clock (scripts/ubench/valu_mix.cpp, profiles/r03_valu_mix.txt): v_fma_f64 5.4, v_mul_f64 5.0, v_add_f64 5.1, v_max_f64 4.9,
    # v_cndmask_b32 4.0, v_add_u32 3.0, v_mov_b32 2.3, v_rcp / v_rsq_f64 16.8; weighted with the static instruction mix of the MHD sweep
    # (75 % fp64 arithmetic, 25 % 32-bit selects / integer) 4.7 per non-transcendental instruction
    measured = 4.7 * (insts - trans) + 16.8 * trans
    return {"valu_wave_insts_per_launch": insts, "trans_f64_wave_insts_per_launch": trans, "issue_cycles": cycles,
            "capacity_cycles": cap, "frac": cycles / cap, "clock_ghz": 2.4, "simds": 1024,
            "frac_of_measured_issue_rate": measured / cap,
            "note": "frac: share of the nominal fp64 VALU issue slots (4 cycles per wave instruction, 16 for rcp/rsq, 2.4 GHz) the kernel fills at its "
                    "measured duration; frac_of_measured_issue_rate: the same against the rates instruction streams of the kernel's mix reach at 2 waves "
                    "per SIMD (4.7 cycles per instruction, 16.8 per rcp/rsq: profiles/r03_valu_mix.txt), averaged over all SIMDs -- in the exact MHD "
                    "sweep the three SIMDs of the Riemann waves are the critical path (~95 % busy) and the producer pair's SIMD carries ~72 % of their "
                    "load; the contracted one is balanced (profiles/r06_sweep_isa_mix.txt, DESIGN.md section 3.1.1)"}


class Control:
    """torch.distributed as the CONTROL plane of a multi-rank run: the 128-byte RCCL id, the barriers of the timing contract,
    the max over ranks of the wall time.  Backend "nccl" (= RCCL; the driver's launch) or, RGPU_BENCH_BACKEND=gloo, gloo -- the
    data plane (halo planes, 1/dt) is the C++ slab driver's own RCCL communicator either way."""

    def __init__(self, world, rank, local_rank):
        import torch
        self.torch, self.world, self.rank = torch, world, rank
        self.dist = None
        self.backend = None
        if world > 1:
            import torch.distributed as dist
            self.dist = dist
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            # Control plane = gloo by default (round 5): it carries 128 bytes, a few python objects and the barriers around the timed region,
            # all on the host; the ONE RCCL communicator of the process is then the slab driver's own (a second one -- torch's NCCL process
            # group -- would share the device's CUs and queues with it for no benefit, and that pairing has never run anywhere).
            # RGPU_BENCH_BACKEND=nccl selects torch's RCCL group instead; if gloo cannot be set up it is the fallback.
            self.backend = os.environ.get("RGPU_BENCH_BACKEND", "gloo")
            if self.backend == "gloo":
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # one node (the contract's launch line): the loopback interface
                try:
                    dist.init_process_group("gloo")
                except Exception as e:  # noqa: BLE001
                    sys.stderr.write("bench.py: gloo control plane failed (%r), using torch's nccl group\n" % (e,))
                    self.backend = "nccl"
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            elif self.backend != "gloo":
                dist.init_process_group(self.backend)

    def _dev(self):
        return "cuda" if self.backend == "nccl" else "cpu"

    def sync(self):
        self.torch.cuda.synchronize()
        if self.dist:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def max(self, x):
        if not self.dist:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def min_int(self, x):
        if not self.dist:
            return x
        t = self.torch.tensor([x], dtype=self.torch.int64, device=self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return int(t.item())

    def bcast(self, obj):
        if not self.dist:
            return obj
        box = [obj]
        self.dist.broadcast_object_list(box, src=0)
        return box[0]

    def gather(self, obj):
        if not self.dist:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def close(self):
        if self.dist:
            self.dist.barrier()
            self.dist.destroy_process_group()


def device_facts(torch, local_rank):
    pr = torch.cuda.get_device_properties(local_rank)
    pci = None
    if all(hasattr(pr, a) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
        pci = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    return {"device": local_rank, "name": pr.name, "pci_bus_id": pci}


def fingerprint_record(dts, checksums):
    """What a run computed, in a form that does not depend on the number of ranks: sha256 of the dt sequence (little-endian doubles of
    all W + K steps) and the sum mod 2^64 of the ranks' state checksums (rgpu_state_checksum: the 64-bit patterns of every interior
    double, an order-independent sum).  The slab driver is bit-identical to the single-device run by construction, so every N must
    print the same record for the same arithmetic, box, W and K -- the driver's N = 1 and N = 8 lines are comparable."""
    import hashlib
    import struct
    return {"dt_sha256": hashlib.sha256(b"".join(struct.pack("<d", float(d)) for d in dts)).hexdigest(),
            "state_sum_u64": "%016x" % (sum(int(c) for c in checksums) % (1 << 64)), "steps": len(dts),
            "of": "state after warmup + steps steps, interior cells; equal for every --gpus N at the same arithmetic, box, steps and warmup"}


def timed_steps(step, timers_src, ctl, steps, warmup, batch=None, dts=None):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; returns the max over ranks of the wall
    time of the K steps [s].  (Per-kernel durations are a separate pass: phase_profile.)  batch(K): the K steps as ONE call of the
    product's loop body (rgpu_run_steps: K turns of the reference's time loop, the same states and dt sequence; where a step is one
    fused kernel the time step stays on the device between steps) -- single-device runs."""
    log = dts if dts is not None else []      # the time step of every step, warmup included (fingerprint_record)
    for _ in range(warmup):
        log.append(step())
    timers_src.enable_timers(False)
    ctl.sync()
    t0 = time.perf_counter()
    if batch is not None:
        done = batch(steps)
        assert done == steps, "rgpu_run_steps did %d of %d steps" % (done, steps)
    else:
        for _ in range(steps):
            log.append(step())
    ctl.sync()
    if batch is not None:
        log.extend(timers_src.dt_log)
    elapsed = ctl.max(time.perf_counter() - t0)
    return elapsed


def phase_profile(step, timers_src, ctl, steps):
    """per-kernel durations: the same steps again with HIP events around every launch on the kernels' stream (separate from
    the timed region: the events serialise host and device)"""
    timers_src.enable_timers(True)
    timers_src.reset_timers()
    nprof = min(5, max(steps, 1))
    for _ in range(nprof):
        step()
    ctl.sync()
    tm = timers_src.timers()
    dom_name, dom_ms, dom_launches = timers_src.dominant_kernel()
    timers_src.enable_timers(False)
    return nprof, tm, dom_name, dom_ms, dom_launches


def roofline_of(wname, w, arith, step_bytes, elapsed, steps, prof):
    nprof, tm, dom_name, dom_ms, dom_launches = prof
    # per step: a slab run launches the kernel once per plane range (two boundary ranges + the inner one)
    dom_ms = dom_ms * dom_launches / nprof
    achieved = step_bytes / (dom_ms * 1e-3)
    pkey = wname if arith == "exact" else wname + "_contracted"   # key of the committed PMC summary (profiles/pmc_traffic.json)
    # the committed counters are per launch of the workload's DEFAULT box (what scripts/prof_round.sh runs): another box, or a slab of
    # it in several launches per step, takes them in proportion to the cells a launch covers
    per_step = max(dom_launches / nprof, 1.0)
    scale = (step_bytes / w["bytes"]) / (float(w["size"][0]) * w["size"][1] * w["size"][2]) / per_step
    vc = valu_ceiling(pkey, dom_name, dom_ms / per_step, scale)
    # what binds the dominant kernel: the 3D MHD sweep is bound by fp64 vector issue ("valu_f64": its share of the issue slots is
    # valu_ceiling.frac), the other sweeps by HBM; `achieved` / `peak` / `frac` stay the contract's HBM figures either way (= hbm_frac)
    bound = "valu_f64" if (w["bytes"] == 128.0 and w["size"][2] > 1 and vc is not None and vc["frac"] > achieved / HBM_PEAK) else "hbm"
    roof = {"bound": bound, "kernel": dom_name, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
            "frac": achieved / HBM_PEAK, "hbm_frac": achieved / HBM_PEAK, "valu_frac": (vc or {}).get("frac"), "traffic": pmc_traffic(pkey, dom_name, False, scale), "traffic_fetch_doubled": pmc_traffic(pkey, dom_name, True, scale),
            "pmc_scale": scale,
            "traffic_note": "HBM bytes per launch from rocprofv3 --pmc (profiles/pmc_traffic.json): FETCH_SIZE + WRITE_SIZE as counted = a lower bound (gfx950 tallies "
                            "128-B read requests at 64 B; 0.49-0.77 of known byte counts on this code's 8 B per lane streams), and with FETCH_SIZE doubled as "
                            "MI355X_MICROARCH.md prescribes for wide streaming reads = an upper bound",
            "algorithmic_bytes_per_launch": step_bytes, "avg_launch_ms": dom_ms, "launches_per_step": dom_launches / nprof, "launches_timed": dom_launches,
            "note": ("fp64-VALU-bound kernel (div / sqrt heavy HLLD + 2D HLLD solvers): see DESIGN.md; avg_launch_ms is the sweep PHASE of a step -- "
                     "mhd3d_sweep_kernel<SPEC, MhTile<16, 8>> (3D MHD: + its short second launch with the 2 x 32 geometry for the last x face column, "
                     "~0.2 ms at 512^3, + the periodic-layer copy, 0.03 ms)" if w["bytes"] == 128.0 else
                     "LDS-tiled z-marching sweep, one kernel per step: see DESIGN.md"),
            "valu_ceiling": vc}
    if _PMC.get("note"):
        roof["pmc_note"] = _PMC["note"]
    step = {"bound": "hbm", "achieved": step_bytes / (elapsed / steps) / 1e9, "peak": HBM_PEAK / 1e9,
            "unit": "GB/s", "frac": step_bytes / (elapsed / steps) / HBM_PEAK,
            "phase_ms": {k: v / nprof * 1e3 for k, v in tm.items() if v > 0}, "sum_phase_ms": sum(tm.values()) / nprof * 1e3}
    return roof, step


def single_gpu_record(wname, dims, arith, steps, warmup, ctl):
    """one workload on ONE GPU through librgpu.so (exact) or librgpu_fast.so (contracted): value, ms/step, rooflines"""
    from ramsesgpu_amd.solver import Library, Solver, lib_path
    w = WORKLOADS[wname]
    nx, ny, nz = dims
    L = Library(lib_path(arith))
    assert L.arithmetic == arith
    ini = os.path.join(ROOT, "configs", w["base"] + ".ini")
    ov = overrides_for(w, nx, ny, nz)
    p = L.params_from_ini(ini, ov)
    U0 = L.init_condition(ini, ov, p)
    run = Solver(p, L)
    run.upload(U0, both=False)
    del U0
    run.make_all_boundaries(0, 0.0, 0.0)
    # (the reference's h_U.copyTo(h_U2) is not needed: every step writes the whole output array)
    dts = []
    elapsed = timed_steps(run.oneStepIntegration, run, ctl, steps, warmup, batch=run.run_steps, dts=dts)
    fingerprint = fingerprint_record(dts, [run.state_checksum(run.nStep % 2)])
    prof = phase_profile(run.oneStepIntegration, run, ctl, steps)
    run.close()
    cells = float(nx) * ny * nz
    roof, roof_step = roofline_of(wname, w, arith, w["bytes"] * cells, elapsed, steps, prof)
    return {"value": steps * cells / elapsed / 1e6, "unit": "Mcell-updates/s", "ms_per_step": elapsed / steps * 1e3, "steps": steps, "warmup": warmup,
            "roofline": roof, "roofline_step": roof_step, "library": os.path.basename(L.path), "arithmetic": arith, "parity": PARITY[arith],
            "fingerprint": fingerprint}


def slab_driver_run(arith, ini, ov, rank, world, ctl):
    """rank `rank` of `world` z-slabs through the C++ RCCL driver (librgpu_comm[_fast].so) with the initial condition built and its
    ghosts filled; (run, info, None), or (None, None, error) on EVERY rank if any rank failed (the ranks agree through ctl)"""
    from ramsesgpu_amd import comm as rcomm
    from ramsesgpu_amd.solver import Library, lib_path
    err, srun, info = None, None, None
    try:
        L = Library(lib_path(arith))
        if os.environ.get("RGPU_BENCH_DRIVER") == "staged-test":
            # TEST HOOK (tests/test_bench_contract.py): the product's C++ driver compiled against the test-only device-staged transport
            # (tests/emu_dev/rg_transport.h: RCCL's wire replaced by pinned host buffers + gloo), so that the N > 1 line -- schedule,
            # batched time loop, fingerprint -- can be produced by several ranks on ONE GPU and compared with the N = 1 line
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import comm_worker as cw
            CL = rcomm.load_comm_library(os.path.join(ROOT, "tests", "_build", "librgpu_comm_dev%s.so" % ("" if arith == "exact" else "_fast")))
            keep = [cw.EXCHANGE_FN(cw._exchange), cw.ALLREDUCE_FN(cw._allreduce)]
            CL.rgpu_comm_test_set_callbacks(keep[0], keep[1])
            CL._keep_callbacks = keep
        else:
            CL = rcomm.load_comm_library(rcomm.comm_lib_path(arith))
        cid = ctl.bcast(rcomm.unique_id(CL) if rank == 0 else None)
        srun = rcomm.CommRun(ini, ov, rank, world, cid, library=L, comm_library=CL)
    except Exception as e:  # noqa: BLE001 -- reported by the caller, on every rank
        err = e
    if ctl.min_int(1 if srun is not None else 0) == 0:
        if srun is not None:
            srun.close()
        return None, None, err
    info = srun.info()
    # planes must really travel: bytes a rank sends per exchange = (slab interfaces of this rank) x ghostWidth planes x nbVar variables
    # (rounds 2-3 measured a ring of one that exchanged nothing -- DESIGN.md section 6 -- and nothing noticed)
    p = srun.p
    from ramsesgpu_amd import _capi
    faces = int(p.bc[4] == _capi.BC_COPY) + int(p.bc[5] == _capi.BC_COPY)
    want = faces * p.ghostWidth * (p.nx + 2 * p.ghostWidth) * (p.ny + 2 * p.ghostWidth) * p.nbVar * 8
    bad = srun.halo_bytes() != want or (world > 1 and want == 0)
    if ctl.min_int(0 if bad else 1) == 0:
        err = RuntimeError("rank %d sends %d bytes per halo exchange, expected %d" % (rank, srun.halo_bytes(), want)) if bad else None
        srun.close()
        return None, None, err
    srun.init_simulation()
    return srun, info, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="mri")
    ap.add_argument("--size", type=int, default=0, help="box edge (cubic; default: the workload's BASELINE size)")
    ap.add_argument("--nx", type=int, default=0)
    ap.add_argument("--ny", type=int, default=0)
    ap.add_argument("--nz", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--arith", choices=["exact", "contracted"], default=os.environ.get("RGPU_ARITH", "contracted"),
                    help="contracted (default): librgpu_fast.so, the tolerance-grade build gated at north_star's relative L2 < 1e-12; "
                         "exact: librgpu.so, bit-identical to the reference")
    ap.add_argument("--no-second-arith", "--no-contracted", dest="no_second", action="store_true",
                    help="N=1: skip the second measurement with the other build of the library (value_exact / value_tolerance)")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="default N=1 headline run only: skip the short implode3d 256^3 / orszag-tang 512^2 measurements nested under other_workloads")
    ap.add_argument("--timeline-only", action="store_true", help="stop after the timed region (for rocprofv3 --kernel-trace concurrency analysis)")
    args = ap.parse_args()

    w = WORKLOADS[args.workload]
    nx, ny, nz = w["size"]
    custom_size = bool(args.size or args.nx or args.ny or args.nz)
    if args.size:
        nx, ny, nz = args.size, args.size, (args.size if nz != 1 else 1)
    nx, ny = args.nx or nx, args.ny or ny
    if nz != 1:
        nz = args.nz or nz
    two_d = nz == 1

    import torch
    from ramsesgpu_amd.solver import Library, Solver, lib_path

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
    # test hook (not used by the driver): RGPU_BENCH_ONE_DEVICE=1 puts every rank on GPU 0 (with RGPU_BENCH_BACKEND=gloo for the
    # control plane) -- on a 1-GPU box this exercises the launch line up to RCCL's refusal of two ranks on one device
    if os.environ.get("RGPU_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        # one rank drives one device (the reference: HydroMpiParameters.cpp:196-201)
        sys.stderr.write("bench.py: rank %d is to drive GPU %d but only %d device(s) are visible\n" % (rank, local_rank, torch.cuda.device_count()))
        sys.exit(4)
    torch.cuda.set_device(local_rank)
    ctl = Control(world, rank, local_rank)

    replicas = two_d and world > 1      # 2D boxes do not shard: independent replicas
    cells_box = float(nx) * ny * nz
    cells_global = cells_box * (world if replicas else 1)
    cells_local = cells_global / world
    dims = "%dx%d" % (nx, ny) if two_d else "%dx%dx%d" % (nx, ny, nz)

    clocked_steps = None
    if world == 1:
        if args.timeline_only:
            L = Library(lib_path(args.arith))
            ini = os.path.join(ROOT, "configs", w["base"] + ".ini")
            ov = overrides_for(w, nx, ny, nz)
            p = L.params_from_ini(ini, ov)
            run = Solver(p, L)
            run.upload(L.init_condition(ini, ov, p), both=False)
            run.make_all_boundaries(0, 0.0, 0.0)
            elapsed = timed_steps(run.oneStepIntegration, run, ctl, args.steps, args.warmup)
            print(json.dumps({"ms_per_step": elapsed / args.steps * 1e3}))
            return
        rec = single_gpu_record(args.workload, (nx, ny, nz), args.arith, args.steps, args.warmup, ctl)
        driver, rccl_ranks = "single device: %s, no communicator" % rec["library"], None
        ranks = [dict(device_facts(torch, local_rank), rank=0)]
    else:
        # the C++ z-slab driver (include/rgpu_comm.h): RCCL halo exchange on a side stream, 1/dt all-reduced on the device;
        # torch.distributed only carries the 128-byte unique id and the barriers of the timing contract.  There is NO fallback:
        # if the driver cannot be created on any rank, every rank exits non-zero with the RCCL error.  RGPU_BENCH_DRIVER=python
        # explicitly selects the test harness (tests/slab_harness.py, the same schedule over torch.distributed) instead.
        L = Library(lib_path(args.arith))
        ini = os.path.join(ROOT, "configs", w["base"] + ".ini")
        ov = overrides_for(w, nx, ny, nz)
        want_python = os.environ.get("RGPU_BENCH_DRIVER", "cpp") == "python"
        info = None
        slab_batch = None
        if replicas:
            p = L.params_from_ini(ini, ov)
            srun = Solver(p, L)
            srun.upload(L.init_condition(ini, ov, p), both=False)
            srun.make_all_boundaries(0, 0.0, 0.0)
            step, timers_src = srun.oneStepIntegration, srun
            driver = "independent replicas: %s per rank, no communicator" % os.path.basename(L.path)
        elif want_python:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from slab_harness import SlabRun
            srun = SlabRun(ini, ov, library=L, device="cuda:%d" % local_rank)
            srun.init_simulation()
            step, timers_src = srun.oneStepIntegration, srun.solver
            driver = "TEST HARNESS (RGPU_BENCH_DRIVER=python): tests/slab_harness.py over torch.distributed/%s -- not the product's driver" % ctl.backend
        else:
            srun, info, err = slab_driver_run(args.arith, ini, ov, rank, world, ctl)
            if srun is None:
                sys.stderr.write("bench.py: rank %d/%d on GPU %d: %s\n" % (rank, world, local_rank,
                                 ("the C++ RCCL slab driver could not be created: %r" % (err,)) if err is not None else "another rank failed to create the C++ RCCL slab driver"))
                sys.stderr.write("bench.py: no fallback (set RGPU_BENCH_DRIVER=python for the torch.distributed test harness)\n")
                sys.stderr.flush()
                sys.exit(5)
            step, timers_src = srun.oneStepIntegration, srun.solver
            slab_batch = srun.run_steps      # rgpu_comm_run_steps: the K timed steps as ONE call, the time step on the device between them
            sched = {0: "0 (serial)", 1: "1 (overlap)", 2: "2 (boundary-first)"}.get(srun.schedule(), "?")
            driver = "%sC++ slab driver librgpu_comm%s.so over %s (include/rgpu_comm.h), schedule %s, %s exchange, %.1f MB sent per rank and step" % (
                "TEST TRANSPORT (RGPU_BENCH_DRIVER=staged-test) -- " if os.environ.get("RGPU_BENCH_DRIVER") == "staged-test" else "",
                "" if args.arith == "exact" else "_fast", info["transport"], sched,
                "in-place (one send / recv per variable and face)" if os.environ.get("RGPU_COMM_PACK") == "0" else "packed (one send / recv per peer)", srun.halo_bytes() / 1e6)
        dts = []
        if slab_batch is not None:
            elapsed = timed_steps(step, srun, ctl, args.steps, args.warmup, batch=slab_batch, dts=dts)   # (srun: the dt log of the batch)
        else:
            elapsed = timed_steps(step, timers_src, ctl, args.steps, args.warmup, dts=dts)
        # what was computed, in a form every N must reproduce (fingerprint_record): every rank's checksum of its own planes
        fingerprint = fingerprint_record(dts, ctl.gather(timers_src.state_checksum((args.warmup + args.steps) % 2)))
        # diagnostic of the halo exchange of the LAST timed step on this rank: its duration on the halo stream and the rate that makes
        # of the bytes this rank sent (include/rgpu_comm.h); next to ms_per_step it tells how much of it the schedule hid
        xchg = None
        if info is not None and srun is not None:
            xms = srun.last_exchange_ms()
            if xms > 0:
                xchg = {"exchange_ms": xms, "sent_MB": srun.halo_bytes() / 1e6, "GB_per_s_sent": srun.halo_bytes() / xms / 1e6}
        if args.timeline_only:
            if rank == 0:
                print(json.dumps({"ms_per_step": elapsed / args.steps * 1e3}))
            ctl.close()
            return
        clocked_steps = srun.clocked_steps() if slab_batch is not None else None
        prof = phase_profile(step, timers_src, ctl, args.steps)
        roof, roof_step = roofline_of(args.workload, w, args.arith, w["bytes"] * cells_local, elapsed, args.steps, prof)
        rec = {"value": args.steps * cells_global / elapsed / 1e6, "ms_per_step": elapsed / args.steps * 1e3, "roofline": roof, "roofline_step": roof_step,
               "fingerprint": fingerprint}
        if info is not None and not args.no_second and os.environ.get("RGPU_BENCH_SECOND_ARITH", "1") != "0":
            # the same slabs, same K and W, through the OTHER build of the library (a second RCCL communicator, after the first is
            # destroyed): `value` stays comparable round over round whichever build is the headline
            other = "exact" if args.arith == "contracted" else "contracted"
            srun.close()
            srun2, info2, err2 = slab_driver_run(other, ini, ov, rank, world, ctl)
            if srun2 is None:
                rec["second"] = (other, {"value": None, "error": repr(err2) if err2 is not None else "another rank failed"})
            else:
                dts2 = []
                el2 = timed_steps(srun2.oneStepIntegration, srun2, ctl, args.steps, args.warmup, batch=srun2.run_steps, dts=dts2)
                fp2 = fingerprint_record(dts2, ctl.gather(srun2.solver.state_checksum((args.warmup + args.steps) % 2)))
                rec["second"] = (other, {"value": args.steps * cells_global / el2 / 1e6, "unit": "Mcell-updates/s", "ms_per_step": el2 / args.steps * 1e3, "fingerprint": fp2,
                                         "steps": args.steps, "warmup": args.warmup, "arithmetic": other, "parity": PARITY[other],
                                         "library": "librgpu_comm%s.so + %s" % ("" if other == "exact" else "_fast", os.path.basename(srun2.L.path)),
                                         "rccl_ranks": info2["ranks"]})
                srun2.close()
        mine = dict(device_facts(torch, local_rank), rank=rank)
        if info is not None:
            mine.update(rccl_rank=info["rank"], rccl_ranks=info["ranks"], rccl_device=info["device"], rccl_pci_bus_id=info["pci_bus_id"])
            mine.update(rccl_version=info.get("rccl_version"))
            if xchg is not None:
                mine.update(last_halo_exchange=xchg)
        # this rank's phase times of a step (hipEvents on its compute stream around every launch, 5 steps): on a multi-GPU run they show
        # whether the traffic on the links slows the fp64-bound sweep of a rank down, rank by rank
        mine.update(phase_ms={k: v / prof[0] * 1e3 for k, v in prof[1].items() if v > 0})
        ranks = ctl.gather(mine)
        rccl_ranks = info["ranks"] if info is not None else None
        if info is not None:   # every rank must have seen the same communicator size, and one device each
            sizes = sorted(set(r.get("rccl_ranks") for r in ranks))
            buses = [r.get("rccl_pci_bus_id") for r in ranks]
            if sizes != [world] or (len(set(buses)) != world and os.environ.get("RGPU_BENCH_DRIVER") != "staged-test"):
                if rank == 0:
                    sys.stderr.write("bench.py: RCCL saw communicator sizes %s for WORLD_SIZE=%d, devices %s\n" % (sizes, world, buses))
                ctl.close()
                sys.exit(6)

    if rank == 0:
        out = {
            "metric": "Mcell-updates/s", "value": rec["value"], "unit": "Mcell-updates/s",
            # which build of the library `value` is: since round 3 the tolerance-grade one by default (metric_version 2; rounds 1-2
            # quoted the bit-identical library -- compare those with value_exact, which rides beside `value` at every N)
            "value_arithmetic": args.arith, "metric_version": 2,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": rec["ms_per_step"],
            "higher_is_better": True, "scaling": "weak" if replicas else "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": w["desc"] % dims,
                       "nx": nx, "ny": ny, "nz": nz,
                       "decomposition": ("%d independent replicas" % world) if replicas else "z-slabs x%d" % world,
                       "driver": driver, "rccl_ranks": rccl_ranks, "ranks": ranks,
                       "path": w["path"],
                       "arithmetic": args.arith,
                       "time_loop": ("rgpu_run_steps(K): K turns of the reference's loop body in one call, same states and dt sequence; dt, t += dt and the "
                                     "loop condition stay on the device between the steps (csrc/step_clock_rec.h, hip/step_clock.h), the host reads the records of "
                                     "the batch once" if world == 1 else
                                     ("rgpu_comm_run_steps(K): the same loop body over the slabs -- per step the 1/dt slots all-reduced in place, the clock record, "
                                      "the step pieces and the halo exchange, no host turn in between; %s of the %d timed + warm-up steps took their time step from "
                                      "the device" % (clocked_steps, args.steps + args.warmup) if clocked_steps is not None else "K calls of oneStepIntegration")),
                       "parity": PARITY[args.arith],
                       "fingerprint": rec.get("fingerprint")},
            "roofline": rec["roofline"], "roofline_step": rec["roofline_step"],
        }
        if world > 1 and rec.get("second"):
            out["value_exact" if rec["second"][0] == "exact" else "value_tolerance"] = rec["second"][1]
        if world == 1 and not args.no_second:
            # the same workload, same K and W, through the OTHER build of the library, with its own roofline: the bit-identical one
            # beside the default tolerance-grade headline (value_exact), or the other way round with --arith exact (value_tolerance)
            other = "exact" if args.arith == "contracted" else "contracted"
            key = "value_exact" if other == "exact" else "value_tolerance"
            try:
                out[key] = single_gpu_record(args.workload, (nx, ny, nz), other, args.steps, args.warmup, ctl)
            except Exception as e:  # noqa: BLE001 -- a secondary number: report why it is missing
                out[key] = {"value": None, "error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w, (nx, ny, nz))
            allc = cpu_baseline_all_cores(w, (nx, ny, nz))
            if allc:
                out["cpu_baseline_all_cores"] = allc
        if world == 1 and args.workload == "mri" and not custom_size and not args.no_other_workloads:
            # BASELINE.json configs[1] and configs[2], short, so that their numbers are the driver's too (not builder-only claims)
            others = {}
            for name, st, wu, budget in (("implode3d", 100, 5, 8.0), ("orszag-tang", 400, 10, 6.0)):   # (2D steps take ~0.05 ms: enough of them that the one synchronisation of the batch does not show)
                try:
                    ow = WORKLOADS[name]
                    o = single_gpu_record(name, ow["size"], args.arith, st, wu, ctl)
                    if args.arith != "exact" and not args.no_second:
                        e = single_gpu_record(name, ow["size"], "exact", st, wu, ctl)
                        o["value_exact"] = {k: e[k] for k in ("value", "unit", "ms_per_step", "roofline", "library", "arithmetic", "parity")}
                    o["config"] = {"workload": ow["desc"] % ("%dx%d" % ow["size"][:2] if ow["size"][2] == 1 else "%dx%dx%d" % ow["size"]), "path": ow["path"]}
                    if not args.no_cpu_baseline:
                        o["cpu_baseline"] = cpu_baseline(ow, ow["size"], budget_s=budget)
                    others[name] = o
                except Exception as e:  # noqa: BLE001
                    others[name] = {"value": None, "error": repr(e)}
            out["other_workloads"] = others
        print(json.dumps(out))
    ctl.close()


if __name__ == "__main__":
    main()
