#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing.

  static:   python scripts/isa_mix.py file.s <substring of the kernel's mangled name> [--blocks]
            per instruction class, in total and (--blocks) per basic block

  dynamic:  python scripts/isa_mix.py --sweep fast|exact [--measured SQ_INSTS_VALU] [--iters N]
            compiles csrc/rgpu_api.cpp with the flags of ramsesgpu_amd/build.py + -gline-tables-only into a listing and weights the
            basic blocks of the main loop(s) of mhd3d_sweep_kernel<107, MhTile<16, 8>> by how many waves of the 512-thread
            workgroup execute them per z plane:
              * every instruction carries the source line it was inlined from (.loc); a line belongs to a device function, a
                function to a wave ROLE (riemann / trace / prim / elec / sync; arithmetic helpers are shared);
              * a basic block belongs to the role most of its role-specific instructions come from (the roles are separated by
                wave-uniform branches, so blocks are role-pure; blocks without role-specific lines are loop control);
              * trip counts per plane and workgroup: the compiler emits one copy of the Riemann code per direction, each run by the
                two waves (cell halves) of that direction -> x 2; one copy of the trace per pass (producer 0: two passes, producer
                1: one) -> x 1; primitives: both producers -> x 2; the producer pair's rendezvous x 2; loop control x 8;
                electric field: every Riemann thread computes value e = its number (0..383) and, if < 512, e + 384 of the plane's
                512 (170 Ex, 162 Ey, 180 Ez); the compiler emits one block per (value, component), run by the waves that hold
                such values -> first value: Ex x 3 waves, Ey x 4, Ez x 1; second value: Ez x 2.  Blocks outside the z loop (prologue) are not counted;
              * the 2D HLLD solver evaluates only the regions of the Riemann fan some lane of the wave needs (dev_numerics.h,
                "Region selection by sign bits"): in the shearing box every speed is below the fast speed, all lanes take the
                inner region and the four outer-region blocks are skipped (s_cbranch_execz) -> x 0.  Exact arithmetic: the
                Alfven speeds come from the selection (dev_numerics.h: alfven_pick / alfven_duel); the reference's own sequence of
                twelve roots is the fallback of a wave with an unsure lane -> x 0 (rgpu_selftest_alfven counts those waves: none in
                10^7 smooth states).
            --measured: SQ_INSTS_VALU of one launch (rocprofv3 --pmc, scripts/pmc_ab.sh); --iters: workgroup x plane iterations of
            that launch (default: the 512^3 shearing box, 2048 tiles x 515 iterations) -> the reconciliation line.
"""
import os
import re
import subprocess
import sys
from collections import Counter, OrderedDict, defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VALU_CLASSES = ("fma64", "mul64", "add64", "trans64", "minmax64", "vcmp", "cndmask", "lane", "accvgpr", "vmov", "vint", "vother")


def classify(op):
    if op.startswith(("v_fma_f64", "v_fmac_f64")): return "fma64"
    if op.startswith("v_mul_f64"): return "mul64"
    if op.startswith("v_add_f64"): return "add64"
    if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")): return "trans64"
    if op.startswith(("v_max_f64", "v_min_f64")): return "minmax64"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"): return "vcmp"
    if op.startswith("v_cndmask"): return "cndmask"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "lane"
    if op.startswith("v_accvgpr"): return "accvgpr"
    if op.startswith(("v_mov", "v_pk_mov")): return "vmov"
    if op.startswith(("v_add_u32", "v_add_co", "v_addc", "v_sub", "v_lshl", "v_lshr", "v_ashr", "v_mul_lo", "v_mul_hi", "v_mad_u", "v_mad_i", "v_and", "v_or", "v_xor", "v_bfe", "v_add3", "v_lshl_add", "v_add_lshl", "v_mul_u32", "v_mad_u64", "v_bfi", "v_not", "v_mul_i32", "v_max_u", "v_min_u", "v_max_i", "v_min_i", "v_mbcnt", "v_add_i32", "v_sub_u32", "v_subrev")): return "vint"
    if op.startswith("v_"): return "vother"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "flat_", "buffer_")): return "vmem"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_"): return "salu"
    return "other"


def kernel_lines(lines, key):
    """the lines of the first function whose mangled name contains `key`"""
    start = None
    for i, l in enumerate(lines):
        if l.startswith("_Z") and key in l and re.match(r"^_Z\S+:", l):
            start = i
            break
    if start is None:
        sys.exit("kernel not found: " + key)
    end = start
    for j in range(start + 1, len(lines)):
        if lines[j].startswith(".Lfunc_end"):
            end = j
            break
    return start, end


def parse_blocks(lines, start, end):
    """OrderedDict label -> {"ops": [(class, (file, line))], "in_loop": bool, "loops": set of the headers (BBn_m) of the loops around it}"""
    blocks = OrderedDict()
    cur = "entry"
    blocks[cur] = {"ops": [], "in_loop": False, "loops": set()}
    loc = (0, 0)
    frames = ()

    def loop_notes(text, b, label):
        m = re.search(r"in Loop: Header=(BB\w+)", text)
        if m:
            b["in_loop"] = True
            b["loops"].add(m.group(1))
        m = re.search(r"Parent Loop (BB\w+)", text)
        if m:
            b["loops"].add(m.group(1))
        if "Loop Header" in text:   # "=>This Loop Header: Depth=1" / "=>  This Inner Loop Header"
            b["in_loop"] = True
            b["loops"].add(label.lstrip(".L"))
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\S+):(.*)", l)
        if m:
            cur = m.group(1)
            blocks[cur] = {"ops": [], "in_loop": False, "loops": set()}
            loop_notes(m.group(2), blocks[cur], cur)
            continue
        t = l.strip()
        m = re.match(r"^; %bb\.(\d+):(.*)", t)
        if m:   # a block the previous one falls through into (no label of its own)
            cur = "%bb." + m.group(1)
            blocks[cur] = {"ops": [], "in_loop": False, "loops": set()}
            loop_notes(m.group(2), blocks[cur], cur)
            continue
        m = re.match(r"^\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            loc = (int(m.group(1)), int(m.group(2)))
            # the inline stack of the location, innermost first, as (file name, line): "; a.h:10:3 @[ b.h:20:5 @[ c.h:30:1 ] ]"
            frames = tuple((os.path.basename(f), int(n)) for f, n in re.findall(r"([\w./+-]+\.(?:h|cpp)):(\d+):\d+", t.split(";", 1)[1] if ";" in t else ""))
            continue
        if t.startswith(";") and ("Loop" in t):
            loop_notes(t, blocks[cur], cur)
            continue
        if not t or t.startswith((";", ".", "//")):
            continue
        blocks[cur]["ops"].append((classify(t.split()[0]), loc))
        blocks[cur].setdefault("frames", []).append(frames)
    for _ in range(4):   # a block of an inner loop names its own header only: add the loops around that header
        for b in blocks.values():
            for h in list(b["loops"]):
                hb = blocks.get(".L" + h)
                if hb:
                    b["loops"] |= hb["loops"]
    return blocks


def static_report(path, key, show_blocks):
    lines = open(path).read().splitlines()
    start, end = kernel_lines(lines, key)
    blocks = parse_blocks(lines, start, end)
    meta = {}
    for l in lines[start:]:
        m = re.match(r"^; (NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|SGPRBlocks|NumSgprs|Occupancy|LDSByteSize|codeLenInByte): (\S+)", l)
        if m and m.group(1) not in meta:
            meta[m.group(1)] = m.group(2)
        if len(meta) >= 8:
            break
    tot = Counter()
    for b in blocks.values():
        tot.update(c for c, _ in b["ops"])
    valu = sum(v for k, v in tot.items() if k in VALU_CLASSES)
    print("kernel", key, meta)
    print("total static instructions", sum(tot.values()), " VALU", valu)
    for k, v in tot.most_common():
        print("  %-10s %6d" % (k, v))
    if show_blocks:
        print("blocks with >= 40 instructions:")
        for name, b in blocks.items():
            c = Counter(x for x, _ in b["ops"])
            n = sum(c.values())
            if n >= 40:
                print("  %-14s %5d  %s" % (name, n, dict(c.most_common(8))))


# ---- dynamic mix of the MHD sweep -------------------------------------------------------------------------------------------
ROLE_OF = {}
for fn in ("riemann_dir edge_state3d face_state3d floor3d edge_emf mag_hlld_2d mhd_hlld mhd_face_flux store_flux fast_speed_sq fast_speed mhd_riemann "
           "max_of4 min_of4 pos_max sel_max sel_min alfven_pick alfven_duel rg_recip_sqrt_pos rg_sqrt_radicand get stride mhd_hll mhd_llf mag_hlla_2d mag_hllf_2d mag_llf_2d mag_hll_average").split():
    ROLE_OF[fn] = "riemann"
for fn in "mhd_trace3d_at mhd_trace3d_finish tvd_half_slope tvd_slope positivity_limiter trace_cell put e_ready".split():
    ROLE_OF[fn] = "trace"
for fn in "mhd_prim prim_load prim_compute prim_store".split():
    ROLE_OF[fn] = "prim"
for fn in "mhd_elec_comp elec_plane".split():
    ROLE_OF[fn] = "elec"
ROLE_OF["pair_sync"] = "sync"
for fn in "mhd3d_update_role mhd_update3d_column_at mhd_update3d_apply info_speeds closing_column n_closing rg_slot_max".split():
    ROLE_OF[fn] = "update"   # the update role of the fused launch: other workgroups, not part of the z march
TRIPS = {"riemann": 2, "trace": 1, "prim": 2, "elec": 6, "sync": 2, "control": 8, "riemann_outer": 0}


def refseq_lines():
    """line range of the reference's own Alfven-speed sequence in mag_hlld_2d (the fallback of the selection; exact arithmetic)"""
    src = open(os.path.join(ROOT, "ramsesgpu_amd", "csrc", "dev_numerics.h")).read().splitlines()
    lo = hi = 0
    for n, l in enumerate(src, 1):
        if lo == 0 and "const rg_recip_t iqLL = rg_recip_sqrt_pos(rstarLL)" in l:
            lo = n
        if lo and "const double SAL = fmin(ustar - calfvenL, 0.0);" in l:
            hi = n
            break
    return lo, hi


def outer_region_lines():
    """line range of the four outer-region branches of mag_hlld_2d in csrc/dev_numerics.h"""
    src = open(os.path.join(ROOT, "ramsesgpu_amd", "csrc", "dev_numerics.h")).read().splitlines()
    lo = hi = 0
    for n, l in enumerate(src, 1):
        if lo == 0 and l.strip().startswith("if (SB_pos) {"):
            lo = n
        if lo and l.strip() == "return E;":
            hi = n
            break
    return lo, hi


def function_map(path):
    """line number -> name of the function / lambda whose definition precedes it"""
    out, cur = {}, None
    try:
        src = open(path).read().splitlines()
    except OSError:
        return out
    for n, l in enumerate(src, 1):
        m = re.match(r"^\s*(?:template\s*<[^>]*>\s*)?(?:RG_DEVFN|inline|static|__global__|__device__)\b[^;=]*?\b(\w+)\s*\([^;]*$", l)
        if m and not l.strip().startswith(("return", "if", "for", "while")):
            cur = m.group(1)
        m2 = re.match(r"^\s*auto\s+(\w+)\s*=\s*\[&\]", l)
        if m2:
            cur = m2.group(1)
        out[n] = cur
    return out


def dynamic_report(arith, measured, iters):
    sys.path.insert(0, ROOT)
    from ramsesgpu_amd import build as rb
    out = "/tmp/isa_mix_%s.s" % arith
    src = os.path.join(rb.CSRC, "rgpu_api.cpp")
    cmd = [rb.HIPCC, "--offload-arch=" + rb.ARCH] + rb.COMMON + (rb.FAST_FLAGS if arith == "fast" else []) + \
        ["-gline-tables-only", "-x", "hip", "--offload-device-only", "-S", src, "-o", out]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().splitlines()
    files = {}
    for l in lines:
        m = re.match(r'^\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', l)
        if m:
            d, f = m.group(2), m.group(3)
            files[int(m.group(1))] = f if os.path.isabs(f) else os.path.join(d if os.path.isabs(d) else os.path.join(ROOT, d), f)
    fmaps = {}
    key = "mhd3d_sweep_kernelILi107ENS_6MhTileILi16ELi8ELb0"
    start, end = kernel_lines(lines, key)
    blocks = parse_blocks(lines, start, end)

    def role_of_loc(loc):
        fid, ln = loc
        path = files.get(fid)
        if not path:
            return None
        if path not in fmaps:
            fmaps[path] = function_map(path)
        return ROLE_OF.get(fmaps[path].get(ln))

    olo, ohi = outer_region_lines()
    rlo, rhi = refseq_lines() if arith != "fast" else (0, 0)
    dn = os.path.join(ROOT, "ramsesgpu_amd", "csrc", "dev_numerics.h")
    per_role = defaultdict(Counter)      # role -> class -> wave instructions per plane and workgroup
    static_role = defaultdict(Counter)
    copies = Counter()
    by_function = Counter()
    # the z march: the loop(s) whose blocks hold Riemann / trace code and no update-role code
    per_loop = defaultdict(Counter)
    byname = {}   # source file name -> path (for the inline stacks, which name files by path)
    for pth in files.values():
        byname.setdefault(os.path.basename(pth), pth)

    def role_of_frames(fr):
        """innermost frame of the inline stack whose function belongs to a role"""
        for fname, ln in fr:
            path = byname.get(fname)
            if not path:
                continue
            if path not in fmaps:
                fmaps[path] = function_map(path)
            r = ROLE_OF.get(fmaps[path].get(ln))
            if r:
                return r
        return None
    for name, b in blocks.items():
        frs = b.get("frames", [])
        votes = Counter(r for r in (role_of_frames(fr) if fr else role_of_loc(loc) for (_, loc), fr in zip(b["ops"], frs + [()] * (len(b["ops"]) - len(frs)))) if r)
        b["votes"] = votes
        for h in b["loops"]:
            per_loop[h].update(votes)
    zloops = set(h for h, v in per_loop.items() if (v["riemann"] + v["trace"]) > 0 and v["update"] == 0)
    def block_role(b):
        return b["votes"].most_common(1)[0][0] if b["votes"] else "control"
    # electric field: which component a block computes (source lines of mhd_elec_comp's three branches), which of the thread's two
    # values (order of appearance), and how many waves hold such values
    k3 = os.path.join(ROOT, "ramsesgpu_amd", "csrc", "kernels_mhd3d.h")
    src3 = open(k3).read().splitlines()
    comp_line = {}
    cur = None
    for n, l in enumerate(src3, 1):
        if "RG_DEVFN double mhd_elec_comp(" in l: cur = -1
        elif cur is not None and "if (COMP == 0)" in l: cur = 0
        elif cur is not None and "if (COMP == 1)" in l: cur = 1
        elif cur is not None and l.strip().startswith("// Ez"): cur = 2
        elif cur is not None and l.startswith("}"): cur = None
        if cur is not None and cur >= 0: comp_line[n] = cur
    PX, PY = 17, 9
    NEX, NEY, NE = PX * (PY + 1), (PX + 1) * PY, PX * (PY + 1) + (PX + 1) * PY + (PX + 1) * (PY + 1)
    comp_of = lambda e: 0 if e < NEX else 1 if e < NEX + NEY else 2
    ewaves = {(n, c): sum(1 for w in range(6) if any(w * 64 + l + 384 * n < NE and comp_of(w * 64 + l + 384 * n) == c for l in range(64))) for n in (0, 1) for c in (0, 1, 2)}
    seen_comp = Counter()
    for name, b in blocks.items():
        if not b["ops"] or not (b["loops"] & zloops):
            continue
        votes = b["votes"]
        role = block_role(b)
        trips = TRIPS.get(role, 0)
        if role == "elec":
            cc = Counter(comp_line[ln] for _, (fid, ln) in b["ops"] if os.path.realpath(files.get(fid, "")) == os.path.realpath(k3) and ln in comp_line)
            if cc and sum(1 for c, _ in b["ops"] if c in ("fma64", "mul64", "add64")) >= 8:
                comp = cc.most_common(1)[0][0]
                trips = ewaves[(min(seen_comp[comp], 1), comp)]
                seen_comp[comp] += 1
            else:
                trips = 6   # dispatch on the component, address set-up: every Riemann wave
        if role == "riemann":   # an outer-region block of the 2D HLLD solver?
            inside = sum(1 for fr in b.get("frames", []) if any(f == "dev_numerics.h" and olo <= ln < ohi for f, ln in fr))
            if 2 * inside > len(b["ops"]):
                role = "riemann_outer"
            inref = sum(1 for fr in b.get("frames", []) if any(f == "dev_numerics.h" and rlo <= ln < rhi for f, ln in fr))
            if rhi and 2 * inref > len(b["ops"]):
                role = "riemann_outer"
        c = Counter(x for x, _ in b["ops"])
        if "--functions" in sys.argv and trips:
            # (per source function: the innermost frame, and the outermost numerics function the instruction was inlined through)
            for (cls, loc), fr in zip(b["ops"], b.get("frames", [])):
                if cls not in VALU_CLASSES:
                    continue
                names = []
                for fname, ln in fr:
                    path = byname.get(fname)
                    if path:
                        if path not in fmaps:
                            fmaps[path] = function_map(path)
                        names.append(fmaps[path].get(ln) or "?")
                inner = names[0] if names else "?"
                outer = next((n for n in reversed(names) if n in ROLE_OF), "?")
                by_function[(role, outer, inner)] += trips
        if sum(c.values()) >= 200:
            copies[role] += 1
        for k, v in c.items():
            static_role[role][k] += v
            per_role[role][k] += v * trips
    print("# dynamic instruction mix of mhd3d_sweep_kernel<107, MhTile<16, 8>>, %s arithmetic" % ("contracted" if arith == "fast" else "exact"))
    print("# wave instructions per z plane and 512-thread workgroup = static count of the role's blocks in the z loop x trips (see the header of scripts/isa_mix.py)")
    print("# large (>= 200 instructions) copies found per role:", dict(copies), " (one Riemann wave pair per direction, each solver cut at its wave-uniform branches; riemann_outer = the x 0 blocks: outer regions of the 2D solver, the exact build's cold Alfven fallback; trace 3)")
    classes = list(VALU_CLASSES) + ["lds", "vmem", "salu", "waitcnt", "branch", "barrier"]
    print("%-9s %5s %8s | " % ("role", "trips", "VALU") + " ".join("%8s" % c for c in classes))
    tot = Counter()
    for role in ("riemann", "riemann_outer", "trace", "prim", "elec", "sync", "control"):
        c = per_role[role] if TRIPS[role] else static_role[role]
        valu = sum(c[k] for k in VALU_CLASSES)
        if TRIPS[role]:
            tot.update(c)
        print("%-9s %5s %8d | " % (role if TRIPS[role] else "(outer)", "3-4-1/2" if role == "elec" else TRIPS[role], valu) + " ".join("%8d" % c[k] for k in classes))
    valu = sum(tot[k] for k in VALU_CLASSES)
    print("%-9s %5s %8d | " % ("all", "", valu) + " ".join("%8d" % tot[k] for k in classes))
    f64 = sum(tot[k] for k in ("fma64", "mul64", "add64", "trans64", "minmax64"))
    print("fp64 arithmetic %.1f %% of the VALU instructions (fma %.1f, mul %.1f, add %.1f, min/max %.1f, rcp/rsq %.1f); 32-bit integer %.1f %%, "
          "compares + selects %.1f %%, moves / lane ops / other %.1f %%" %
          (100.0 * f64 / valu, 100.0 * tot["fma64"] / valu, 100.0 * tot["mul64"] / valu, 100.0 * tot["add64"] / valu, 100.0 * tot["minmax64"] / valu,
           100.0 * tot["trans64"] / valu, 100.0 * tot["vint"] / valu, 100.0 * (tot["vcmp"] + tot["cndmask"]) / valu,
           100.0 * (tot["vmov"] + tot["lane"] + tot["vother"] + tot["accvgpr"]) / valu))
    print("per cell (128 cells per plane and workgroup): %.0f VALU thread-instructions" % (valu * 64.0 / 128.0))
    # per-SIMD view: waves w and w + 4 share a SIMD
    r, t, p, e, s, ctl = (sum(static_role[x][k] for k in VALU_CLASSES) for x in ("riemann", "trace", "prim", "elec", "sync", "control"))
    edyn = sum(per_role["elec"][k] for k in VALU_CLASSES)
    print("per SIMD and plane (VALU): Riemann SIMD = 2 x (%.0f + control %.0f) + a third of the electric field (%.0f) = %.0f;  "
          "producer SIMD = 3 x %.0f + 2 x (%.0f + %.0f + control %.0f) = %.0f" %
          (r / 3.0, ctl, edyn / 3.0, 2 * (r / 3.0 + ctl) + edyn / 3.0, t / 3.0, p, s, ctl, t + 2 * (p + s + ctl)))
    if by_function:
        print("VALU wave instructions per plane and workgroup by (role, numerics function it was inlined through, innermost function):")
        for (role, outer, inner), n in by_function.most_common(45):
            print("  %6d  %-8s %-22s %s" % (n, role, outer, inner))
    if measured:
        pred = valu * iters
        print("reconciliation: predicted %.4g wave instructions per launch (%d workgroup x plane iterations), SQ_INSTS_VALU measured %.4g: %+.1f %%" %
              (pred, iters, measured, 100.0 * (pred / measured - 1.0)))


def main():
    if "--sweep" in sys.argv:
        arith = sys.argv[sys.argv.index("--sweep") + 1]
        measured = float(sys.argv[sys.argv.index("--measured") + 1]) if "--measured" in sys.argv else 0.0
        iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 2048 * 515
        dynamic_report(arith, measured, iters)
        return
    static_report(sys.argv[1], sys.argv[2], "--blocks" in sys.argv)


if __name__ == "__main__":
    main()
