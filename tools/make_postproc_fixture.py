#!/usr/bin/env python3
"""Fixture for the region post-processing functions, produced by the COMPILED REFERENCE
(oracle/_ref/libbwa_ref.so): seeded random region vectors -- rich in ties, because the reference's unstable sorts decide
on them -- through its own mem_sort_and_dedup (bwamem.c:395), mem_mark_primary_se (:445), mem_approx_mapq_se (:1023),
mem_pestat (bwamem_pair.c:46) and mem_pair (:177).  Output: tests/golden/postproc_golden.npz (data only).
Run in the build container:  python tools/make_postproc_fixture.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kswlib  # noqa: E402
import reflib  # noqa: E402
import postgen  # noqa: E402


def main():
    L = reflib.lib()
    L.mem_sort_and_dedup.restype = C.c_int
    L.mem_sort_and_dedup.argtypes = [C.c_int, C.c_void_p, C.c_float]
    L.mem_mark_primary_se.restype = None
    L.mem_mark_primary_se.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
    L.mem_approx_mapq_se.restype = C.c_int
    L.mem_approx_mapq_se.argtypes = [C.c_void_p, C.c_void_p]
    L.mem_pair.restype = C.c_int
    out = {}
    for si, kw in enumerate(postgen.OPTION_SETS):
        opt = L.mem_opt_init()
        for k, v in kw.items():
            setattr(opt.contents, k, v)
        rng = np.random.default_rng(9000 + si)
        l_pac = 1_000_000
        vecs = postgen.region_vectors(rng, 350, l_pac)
        flat_in = np.concatenate(vecs)
        offs = np.cumsum([0] + [len(v) for v in vecs])
        # dedup
        ded = []
        for v in vecs:
            a = v.copy()
            n = L.mem_sort_and_dedup(len(a), a.ctypes.data_as(C.c_void_p), C.c_float(opt.contents.mask_level_redun)) if len(a) else 0
            ded.append(a[:n].copy())
        # primary marking + mapq on the de-duplicated vectors
        marked, mapq = [], []
        for i, v in enumerate(ded):
            a = v.copy()
            if len(a):
                L.mem_mark_primary_se(opt, len(a), a.ctypes.data_as(C.c_void_p), C.c_int64(12345 + 7 * i))
            marked.append(a)
            mapq.append(np.array([L.mem_approx_mapq_se(opt, a[k:k + 1].ctypes.data_as(C.c_void_p)) for k in range(len(a))], dtype=np.int32))
        # insert-size statistics and pairing over consecutive vectors taken as mates
        pairs = postgen.paired_vectors(rng, 400, l_pac)
        pflat = np.concatenate(pairs)
        poffs = np.cumsum([0] + [len(v) for v in pairs])
        c_regs = kswlib.regs_to_c(pairs)
        pes = np.zeros(4, dtype=kswlib.PESTAT)
        L.mem_pestat.restype = None
        L.mem_pestat(opt, C.c_int64(l_pac), C.c_int(len(pairs)), c_regs, pes.ctypes.data_as(C.c_void_p))
        pr = np.zeros((len(pairs) // 2, 5), dtype=np.int32)
        for k in range(len(pairs) // 2):
            sub, nsub = C.c_int(0), C.c_int(0)
            z = (C.c_int * 2)(-1, -1)
            o = L.mem_pair(opt, C.c_int64(l_pac), None, pes.ctypes.data_as(C.c_void_p), None, C.byref(c_regs, 2 * k * C.sizeof(kswlib.CAlnregV)),
                           C.c_int(1000 + k), C.byref(sub), C.byref(nsub), z)
            pr[k] = (o, sub.value, nsub.value, z[0], z[1])
        kswlib.regs_from_c(c_regs)
        p = f"s{si}_"
        out[p + "in"], out[p + "in_off"] = flat_in, offs
        out[p + "ded"], out[p + "ded_off"] = np.concatenate(ded), np.cumsum([0] + [len(v) for v in ded])
        out[p + "marked"], out[p + "mapq"] = np.concatenate(marked), np.concatenate(mapq)
        out[p + "pairs"], out[p + "pairs_off"], out[p + "pes"], out[p + "pair_res"] = pflat, poffs, pes, pr
    path = os.path.join(ROOT, "tests", "golden", "postproc_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
