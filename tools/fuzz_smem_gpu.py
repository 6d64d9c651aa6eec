#!/usr/bin/env python3
"""One-off fuzz of bmh_smem_batch on a GPU box: both kernels (BMH_SMEM_KERNEL=conv|loops), ragged / mutated / reverse-
complemented reads cut from the fixture's genome, two seeding option sets, every read against the oracle
(oracle/fmindex_oracle.c).  72 000 reads, 0 mismatches when last run.  Usage: python tools/fuzz_smem_gpu.py"""
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import kswlib
from test_fmindex_cpu import _same_calls
from __graft_entry__ import load_package
pkg = load_package()
cb, keep, raw, opt, reads, per, sa_k, sa_pos = kswlib.golden_fmindex()
src = np.concatenate(reads)
bad = 0
for kernel in ("conv", "loops"):
    os.environ["BMH_SMEM_KERNEL"] = kernel
    for seed in (1, 2, 3):
        rng = np.random.default_rng(1000 + seed)
        more = []
        for _ in range(6000):
            L = int(rng.choice([0, 1, 2, 18, 19, 20, 37, 75, 100, 150, 151, 250, 300, 600]))
            if L == 0:
                more.append(np.zeros(0, np.uint8)); continue
            p = int(rng.integers(0, len(src) - L))
            rd = src[p:p + L].copy()
            m = rng.random(L) < float(rng.choice([0.0, 0.01, 0.05, 0.2]))
            rd[m] = (rd[m] + rng.integers(1, 4, m.sum())) % 5
            if rng.random() < 0.3:
                rd = (3 - rd[::-1]) % 5 if (rd < 4).all() else rd[::-1].copy()
            more.append(rd)
        so = dict(opt) if isinstance(opt, dict) else opt
        for variant in range(2):
            o2 = opt.copy()
            if variant == 1:
                o2["min_seed_len"], o2["split_len"], o2["split_width"], o2["start_width"] = 12, 20, 30, 2
            ctx = pkg.Context(0, kswlib.make_params())
            ctx.set_bwt(*raw)
            got = ctx.smem_batch(o2, more)
            for r, (g, rd) in enumerate(zip(got, more)):
                w = kswlib.orc_smem_calls(cb, o2, rd) if len(rd) else (np.zeros(0, kswlib.SMEM_CALL), np.zeros(0, kswlib.SMEM_INTV))
                if not _same_calls(g, w):
                    bad += 1
                    if bad < 5: print("MISMATCH", kernel, seed, variant, r, len(rd))
            ctx.close()
        print(kernel, seed, "done", flush=True)
print("mismatches:", bad)
