#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag> (scripts/prof_round.sh) into the files committed under profiles/:
   <tag>_bench.json, <tag>_kernel_stats_default.csv, <tag>_kernel_stats_serial.csv, <tag>_pmc_per_kernel.csv,
   pmc_traffic.json (FETCH_SIZE + WRITE_SIZE per launch, read by bench.py for roofline.traffic)."""
import csv, glob, json, os, re, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01b"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")

PHASE = [("K_mhd_invdt", "dt"), ("K_mhd_prim", "prim"), ("K_mhd_elec", "elec"), ("K_mhd_trace3d", "trace"),
         ("K_mhd_flux3d<63", "flux"), ("K_mhd_flux3d<7>", "flux"), ("K_mhd_flux3d<56>", "emf"), ("K_mhd_update3d", "update"),
         ("K_shear_save_emf", "shear"), ("K_shear_remap", "shear"), ("K_shear_ghost", "boundaries"), ("K_bc_face", "boundaries")]


def phase_of(kernel):
    for key, ph in PHASE:
        if key in kernel:
            return ph
    return None


line = [l for l in open(os.path.join(src, "bench.json")) if l.startswith("{")][-1]
json.dump(json.loads(line), open(os.path.join(dst, tag + "_bench.json"), "w"), indent=1)
for kind in ("default", "serial"):
    f = glob.glob(os.path.join(src, "trace_" + kind, "**", "*kernel_stats.csv"), recursive=True)
    if f:
        shutil.copy(f[0], os.path.join(dst, "%s_kernel_stats_%s.csv" % (tag, kind)))

# per-kernel PMC averages (the bench's whole-domain launches only: grid >= 1e6 work-items)
acc = defaultdict(lambda: defaultdict(list))
for grp in ("pmc_sq", "pmc_fetch", "pmc_write"):
    for f in glob.glob(os.path.join(src, grp, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            ph = phase_of(row["Kernel_Name"])
            if ph is None or int(row["Grid_Size"]) < 500000:
                continue
            acc[ph][row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
counters = sorted({c for ph in acc for c in acc[ph]})
with open(os.path.join(dst, tag + "_pmc_per_kernel.csv"), "w") as out:
    out.write("phase(kernel),launches_sampled," + ",".join(counters) + "\n")
    traffic = {}
    for ph in ["dt", "prim", "elec", "trace", "flux", "emf", "update"]:
        if ph not in acc:
            continue
        vals = []
        n = 0
        for c in counters:
            # a counter value is reported once per dispatch (summed over XCDs by rocprofv3's csv): average the dispatches
            per = defaultdict(float)
            for d, v in acc[ph].get(c, []):
                per[d] += v
            n = max(n, len(per))
            vals.append(sum(per.values()) / len(per) if per else float("nan"))
        out.write("%s,%d,%s\n" % (ph, n, ",".join("%.6g" % v for v in vals)))
        m = dict(zip(counters, vals))
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            traffic[ph] = {"hbm_bytes_per_launch": (m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0,
                           "fetch_bytes": m["FETCH_SIZE"] * 1024.0, "write_bytes": m["WRITE_SIZE"] * 1024.0,
                           "note": "512^3 MRI, rocprofv3 --pmc, separate passes, serial schedule (one whole-domain launch per kernel); "
                                   "FETCH_SIZE/WRITE_SIZE in KiB x1024. WRITE_SIZE matches the byte count of the stores exactly; "
                                   "FETCH_SIZE is a LOWER bound on gfx950 (128-B requests tallied as 64 B, MI355X_MICROARCH.md HBM "
                                   "section): measured 0.62-0.78x of the known unique bytes on the streaming kernels (prim, dt)"}
json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print("wrote", sorted(os.listdir(dst)))
