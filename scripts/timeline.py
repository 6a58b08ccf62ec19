#!/usr/bin/env python3
"""Concurrency analysis of a rocprofv3 --kernel-trace csv: GPU busy fraction, time-weighted number of kernels in
flight, and which kernel types run together.  usage: timeline.py <kernel_trace.csv> [skip_fraction]"""
import csv, re, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
def short(n):
    m = re.search(r"K_\w+(<[^>]*>)?", n) or re.search(r"(mhd3d|hydro3d)_sweep_kernel", n) or re.search(r"nccl\w*|rccl\w*", n) or re.search(r"emulated_link_hold|step_clock_kernel", n)
    return m.group(0) if m else n[:30]
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows]
ks.sort()
t_lo, t_hi = ks[0][0], max(k[1] for k in ks)
skip = float(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else 0.5
t0 = t_lo + (t_hi - t_lo) * skip            # analyse the tail (timed steps), not initialisation
for s, e, n in ks:
    if e <= t0: continue
    ev.append((max(s, t0), 1, n)); ev.append((e, -1, n))
ev.sort()
active = defaultdict(int); nact = 0; last = ev[0][0]
hist = defaultdict(float); combo = defaultdict(float)
for t, d, n in ev:
    dt = t - last
    if dt > 0:
        hist[nact] += dt
        combo[tuple(sorted(k for k, v in active.items() if v > 0))] += dt
    active[n] += d; nact += d; last = t
tot = sum(hist.values())
print("window %.1f ms" % (tot / 1e6))
for k in sorted(hist): print("  %d kernels in flight: %5.1f %%" % (k, 100 * hist[k] / tot))
print("top combinations:")
for c, v in sorted(combo.items(), key=lambda x: -x[1])[:14]: print("  %5.1f %%  %s" % (100 * v / tot, " + ".join(c) if c else "(idle)"))

if "--gantt" in sys.argv:   # the last launches, one per line, with start .. end relative to the first of them (about two steps of a slab run)
    tail = ks[-56:]
    base = tail[0][0]
    print("the last %d launches (start .. end in ms):" % len(tail))
    for s_, e_, n_ in tail:
        print("  %8.3f .. %8.3f  %s" % ((s_ - base) / 1e6, (e_ - base) / 1e6, n_))
