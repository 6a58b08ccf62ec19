#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag> (scripts/prof_round.sh) into the files committed under profiles/:
   <tag>_bench.json (+ _implode3d, _orszag-tang), <tag>_kernel_stats_<workload>.csv, <tag>_pmc_per_kernel_<workload>.csv,
   pmc_traffic.json (FETCH_SIZE + WRITE_SIZE per launch and workload, read by bench.py for roofline.traffic)."""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02a"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")

PHASE = [("MhTile<2, 32", "sweep_lastx"),   # the short launch of the sweep kernel for the last x face column (hip/tiled_mhd.h: MhLastX)
         ("mhd3d_sweep_kernel", "sweep"), ("hydro3d_sweep_kernel", "sweep"), ("mhd2d_step_kernel", "sweep"), ("K_mhd_invdt", "dt"), ("K_hydro_invdt", "dt"),
         ("K_mhd_prim", "prim"), ("K_mhd_elec", "elec"), ("K_mhd_trace3d", "trace"), ("K_mhd_flux3d", "flux"),
         ("K_mhd_update3d", "update"), ("K_shear_save_emf", "shear"), ("K_shear_remap", "shear"), ("K_shear_ghost", "boundaries"),
         ("K_fill_xy", "boundaries"), ("K_copy_periodic_layer", "sweep_copy"), ("step_clock_kernel", "clock"),
         ("K_bc_face", "boundaries"), ("K_copy_cells", "boundaries"), ("K_hydro_prim", "prim"), ("K_hydro_trace", "trace"),
         ("K_hydro_flux", "flux"), ("K_hydro_update", "update")]
CELLS = {"mri": 512.0 ** 3, "implode3d": 256.0 ** 3, "orszag-tang": 512.0 ** 2}
CELLS.update({k + "_contracted": v for k, v in list(CELLS.items())})


def phase_of(kernel):
    for key, ph in PHASE:
        if key in kernel:
            return ph
    return None


for name in ("bench", "bench_implode3d", "bench_orszag-tang"):
    path = os.path.join(src, name + ".json")
    if os.path.exists(path):
        lines = [l for l in open(path) if l.startswith("{")]
        if lines:
            json.dump(json.loads(lines[-1]), open(os.path.join(dst, "%s_%s.json" % (tag, name)), "w"), indent=1)

traffic = {}
for w in ("mri", "implode3d", "orszag-tang", "mri_contracted", "implode3d_contracted", "orszag-tang_contracted"):
    f = glob.glob(os.path.join(src, "trace_" + w, "**", "*kernel_stats.csv"), recursive=True)
    if f:
        shutil.copy(f[0], os.path.join(dst, "%s_kernel_stats_%s.csv" % (tag, w)))
    # per-kernel PMC averages over the whole-domain launches (small boundary launches excluded by grid size)
    acc = defaultdict(lambda: defaultdict(list))
    for grp in ("pmc_sq_", "pmc_fetch_", "pmc_write_"):
        for f in glob.glob(os.path.join(src, grp + w, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                ph = phase_of(row["Kernel_Name"])
                if ph is None or ph in ("boundaries", "shear") or int(row["Grid_Size"]) < (100000 if not w.startswith("orszag-tang") else 50000):
                    continue
                acc[ph][row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    if not acc:
        continue
    counters = sorted({c for ph in acc for c in acc[ph]})
    traffic[w] = {}
    with open(os.path.join(dst, "%s_pmc_per_kernel_%s.csv" % (tag, w)), "w") as out:
        out.write("phase(kernel),launches_sampled," + ",".join(counters) + ",VALU_inst_per_cell,HBM_bytes_per_cell\n")
        for ph in ["dt", "prim", "elec", "trace", "flux", "sweep", "sweep_lastx", "update"]:
            if ph not in acc:
                continue
            vals, n = [], 0
            for c in counters:
                per = defaultdict(float)   # a counter is reported per dispatch and XCD: sum the XCDs, average the dispatches
                for d, v in acc[ph].get(c, []):
                    per[d] += v
                n = max(n, len(per))
                vals.append(sum(per.values()) / len(per) if per else float("nan"))
            m = dict(zip(counters, vals))
            valu = m.get("SQ_INSTS_VALU", float("nan")) * 64.0 / CELLS[w]
            hbm = (m.get("FETCH_SIZE", float("nan")) + m.get("WRITE_SIZE", float("nan"))) * 1024.0 / CELLS[w]
            out.write("%s,%d,%s,%.1f,%.1f\n" % (ph, n, ",".join("%.6g" % v for v in vals), valu, hbm))
            if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
                traffic[w][ph] = {"hbm_bytes_per_launch": (m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0,
                                  "fetch_bytes": m["FETCH_SIZE"] * 1024.0, "write_bytes": m["WRITE_SIZE"] * 1024.0,
                                  "launches_sampled": n}
                if "SQ_INSTS_VALU" in m:   # wave-level instruction counts of one launch (for the fp64-VALU ceiling in bench.py)
                    traffic[w][ph]["valu_wave_insts"] = m["SQ_INSTS_VALU"]
                    traffic[w][ph]["valu_trans_f64_wave_insts"] = m.get("SQ_INSTS_VALU_TRANS_F64", 0.0)
    # kernels launched every step (a phase sampled far less often -- the one CFL scan of the initial state -- is not one)
    nmax = max([v["launches_sampled"] for k, v in traffic[w].items() if not k.startswith("_")] or [0])
    traffic[w]["_step_total_bytes"] = sum(v["hbm_bytes_per_launch"] for k, v in traffic[w].items()
                                          if not k.startswith("_") and 2 * v["launches_sampled"] >= nmax)
traffic["_note"] = ("rocprofv3 --pmc, separate passes (FETCH_SIZE | WRITE_SIZE), one whole-domain launch per kernel and step; KiB x 1024. "
                    "WRITE_SIZE matches the byte count of the stores; FETCH_SIZE is a LOWER bound on gfx950 (128-B requests tallied as "
                    "64 B, MI355X_MICROARCH.md HBM section, which prescribes doubling it for 16 B per lane streaming reads): 0.49-0.77x of the known unique bytes on this code's 8 B per lane SoA streams "
                    "(the CFL scan of the initial state reads the whole state array once: 5.32 of 8.90 GB at 518^3 x 8, 0.34 of 0.70 GB at 260^3 x 5); bench.py reports FETCH + WRITE (lower bound) and 2 x FETCH + WRITE (upper bound). "
                    "mri = 512^3 MRI box, implode3d = 256^3 hydro implosion (HLLC)")
sha = os.path.join(src, "kernel_source_sha.txt")
if os.path.exists(sha):   # the state of the kernel sources the counters were collected on (ramsesgpu_amd/build.py: kernel_source_hash)
    traffic["_kernel_source_sha"] = open(sha).read().strip()
json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print("wrote", sorted(os.listdir(dst)))
