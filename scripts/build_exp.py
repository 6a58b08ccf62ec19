#!/usr/bin/env python3
"""experiment builds of the product library: build_exp.py tag=flag,flag ...  ->  ramsesgpu_amd/librgpu_exp_<tag>.so"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ramsesgpu_amd import build as rb
from concurrent.futures import ThreadPoolExecutor
jobs = []
for a in sys.argv[1:]:
    tag, flags = a.split("=", 1)
    jobs.append((tag, [f for f in flags.split(",") if f]))
def one(j):
    tag, flags = j
    ef = []
    for f in flags:
        ef += (["-mllvm", f[6:]] if f.startswith("mllvm:") else [f])
    return rb.build(verbose=False, force=True, extra_flags=ef, out_name="librgpu_exp_%s.so" % tag)
with ThreadPoolExecutor(max_workers=4) as ex:
    for r in ex.map(one, jobs):
        print(r)
