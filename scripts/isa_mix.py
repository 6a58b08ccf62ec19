#!/usr/bin/env python3
"""Static instruction mix of one kernel in a hipcc -S listing: per basic block and in total.
   python scripts/isa_mix.py file.s <substring of the kernel's mangled name> [--blocks]"""
import re
import sys
from collections import Counter, OrderedDict


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


def main():
    path, key = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    lines = open(path).read().splitlines()
    start = None
    for i, l in enumerate(lines):
        if l.startswith("_Z") and key in l and l.rstrip().endswith(("", )) and ":" in l and not l.startswith("\t"):
            if re.match(r"^_Z\S+:", l):
                start = i
                break
    if start is None:
        sys.exit("kernel not found")
    blocks = OrderedDict()
    cur = "entry"
    blocks[cur] = Counter()
    meta = {}
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\S+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = Counter()
            continue
        t = l.strip()
        if not t or t.startswith((";", ".", "//")):
            continue
        op = t.split()[0]
        blocks[cur][classify(op)] += 1
    for l in lines[start:]:
        m = re.match(r"^; (NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|SGPRBlocks|NumSgprs|Occupancy|LDSByteSize|codeLenInByte): (\S+)", l)
        if m and m.group(1) not in meta:
            meta[m.group(1)] = m.group(2)
        if l.startswith(".Lfunc_end") and meta.get("Occupancy"):
            pass
        if len(meta) >= 8:
            break
    # spill counts from the .amdhsa / metadata block
    txt = "\n".join(lines)
    tot = Counter()
    for c in blocks.values():
        tot.update(c)
    valu = sum(v for k, v in tot.items() if k in ("fma64", "mul64", "add64", "trans64", "minmax64", "vcmp", "cndmask", "lane", "accvgpr", "vmov", "vint", "vother"))
    print("kernel", key, meta)
    print("total static instructions", sum(tot.values()), " VALU", valu)
    for k, v in tot.most_common():
        print("  %-10s %6d" % (k, v))
    if show_blocks:
        print("blocks with >= 40 instructions:")
        for b, c in blocks.items():
            n = sum(c.values())
            if n >= 40:
                print("  %-14s %5d  %s" % (b, n, dict(c.most_common(8))))


if __name__ == "__main__":
    main()
