#!/usr/bin/env python3
"""Seeding queries on the GPU versus the reference's own code on the host cores (informational; bench.py is the
contract metric).  Needs oracle/_ref (the reference compiled by oracle/Makefile builds the index and is the baseline).

  index   : synthetic genome, `bwa index` of the compiled reference
  reads   : 150 bp, 2 % substitutions + rare indels, half reverse-complemented
  GPU     : bmh_smem_batch (all bwt_smem1 calls of smem_next2's iteration per read) and bmh_sa_batch over the
            occurrences mem_insert_seed would look up; kernel time from HIP events, wall time of the whole call
            (host buffers in and out, reordering) beside it
  CPU     : the reference's smem_itr / smem_next2 loop and bwt_sa on all host threads (oracle/ref_seed_shim.c)
Usage: python tools/fmindex_bench.py [--genome 20000000] [--reads 500000]
"""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kswgen  # noqa: E402
import kswlib  # noqa: E402
import reflib  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=int, default=20_000_000)
    ap.add_argument("--reads", type=int, default=500_000)
    a = ap.parse_args()
    pkg = load_package()
    rng = np.random.default_rng(20261010)
    tmp = tempfile.mkdtemp(prefix="bmh_fmb_")
    ref = kswgen.rand_seq(rng, a.genome)
    fa = os.path.join(tmp, "ref.fa")
    reflib.write_fasta(fa, "synth", ref)
    t0 = time.time()
    reflib.build_index(fa)
    t_index = time.time() - t0
    idx = reflib.lib().bwa_idx_load(fa.encode(), 7)
    prim, L2, sl, words, sai, sa = reflib.bwt_arrays(idx)
    opt = reflib.opt_from_params(kswlib.make_params())
    so = reflib.smem_opt_of(opt)
    L = 150
    pos = rng.integers(0, a.genome - L - 8, size=a.reads)
    reads = ref[pos[:, None] + np.arange(L)[None, :]]
    sub = rng.random(reads.shape) < 0.02
    reads = np.where(sub, (reads + rng.integers(1, 4, reads.shape)) & 3, reads).astype(np.uint8)
    rc = rng.random(a.reads) < 0.5
    reads[rc] = 3 - reads[rc][:, ::-1]
    rlist = list(reads)

    ctx = pkg.Context(0, kswlib.make_params())
    ctx.set_bwt(prim, L2, sl, words, sai, sa)
    ctx.set_kernel_timing(True)
    ctx.smem_batch(so, rlist[:2000])  # warm-up
    t0 = time.perf_counter()
    got = ctx.smem_batch(so, rlist)
    wall_smem = time.perf_counter() - t0
    k_smem = ctx.last_kernel_ms()
    n_calls = sum(len(c) for c, _ in got)
    intv = np.concatenate([iv for _, iv in got])
    slen = (intv["info"] & 0xffffffff).astype(np.int64) - (intv["info"] >> 32).astype(np.int64)
    sel = (slen >= int(so["min_seed_len"])) & (intv["x2"] <= opt.contents.max_occ)
    ks = np.concatenate([np.arange(int(x0), int(x0) + int(x2), dtype=np.uint64) for x0, x2 in zip(intv["x0"][sel], intv["x2"][sel])]) \
        if sel.any() else np.zeros(0, np.uint64)
    ctx.sa_batch(ks[:1000])
    t0 = time.perf_counter()
    posg = ctx.sa_batch(ks)
    wall_sa = time.perf_counter() - t0
    k_sa = ctx.last_kernel_ms()

    # parity on a sample + work count (bwt_extend calls) from the oracle
    keep = []
    cb = kswlib.make_cbwt(prim, L2, sl, words, sai, sa, keep)
    orc = kswlib.load_oracle()
    orc.orc_fm_extends.restype = C.c_uint64
    orc.orc_fm_extends(1)
    ns = min(a.reads, 3000)
    ok = True
    for r in range(ns):
        wc, wi = kswlib.orc_smem_calls(cb, so, rlist[r])
        gc, gi = got[r]
        ok = ok and len(gc) == len(wc) and len(gi) == len(wi) and bool((gi == wi).all()) and bool((gc["ret"] == wc["ret"]).all())
    ext_per_read = orc.orc_fm_extends(1) / ns
    ok_sa = bool((posg[:5000] == kswlib.orc_sa(cb, ks[:5000])).all())

    # CPU reference on all host threads
    shim = C.CDLL(os.path.join(kswlib.REF_DIR, "libref_seed_shim.so"))
    shim.ref_smem_iter_mt.restype = C.c_uint64
    ncores = os.cpu_count() or 1
    pool = np.ascontiguousarray(reads.reshape(-1))
    off = (np.arange(a.reads, dtype=np.uint64) * L)
    lens = np.full(a.reads, L, dtype=np.int32)
    cs = C.c_uint64(0)
    bwt_p = C.c_void_p(idx.contents.bwt)
    args = (bwt_p, C.c_int(a.reads), pool.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p),
            lens.ctypes.data_as(C.c_void_p), C.c_int(int(so["split_len"])), C.c_int(int(so["split_width"])),
            C.c_int(int(so["start_width"])), C.c_int(ncores), C.byref(cs))
    shim.ref_smem_iter_mt(*args)
    t0 = time.perf_counter()
    shim.ref_smem_iter_mt(*args)
    cpu_smem = time.perf_counter() - t0
    posc = np.zeros(len(ks), dtype=np.uint64)
    shim.ref_sa_mt(bwt_p, ks.ctypes.data_as(C.c_void_p), C.c_int(len(ks)), posc.ctypes.data_as(C.c_void_p), C.c_int(ncores))
    t0 = time.perf_counter()
    shim.ref_sa_mt(bwt_p, ks.ctypes.data_as(C.c_void_p), C.c_int(len(ks)), posc.ctypes.data_as(C.c_void_p), C.c_int(ncores))
    cpu_sa = time.perf_counter() - t0
    ok_sa = ok_sa and bool((posc == posg).all())
    blk = 64.0  # one 64-byte index block per occurrence query; two queries per bwt_extend
    out = {"genome_bp": a.genome, "index_bytes": int(words.nbytes + sa.nbytes), "index_s": t_index, "reads": a.reads, "read_len": L,
           "smem": {"calls": n_calls, "intervals": int(len(intv)), "bwt_extend_per_read": ext_per_read, "kernel_ms": k_smem,
                    "wall_ms": wall_smem * 1e3, "reads_per_s_kernel": a.reads / (k_smem * 1e-3),
                    "reads_per_s_wall": a.reads / wall_smem,
                    "algorithmic_GBps_kernel": ext_per_read * 2 * blk * a.reads / (k_smem * 1e-3) / 1e9,
                    "cpu_reference_reads_per_s": a.reads / cpu_smem, "cpu_threads": ncores,
                    "parity": "bit-exact vs oracle (%d reads)" % ns if ok else "MISMATCH"},
           "sa": {"lookups": int(len(ks)), "kernel_ms": k_sa, "wall_ms": wall_sa * 1e3, "lookups_per_s_kernel": len(ks) / (k_sa * 1e-3),
                  "cpu_reference_lookups_per_s": len(ks) / cpu_sa, "parity": "bit-exact vs reference (all)" if ok_sa else "MISMATCH"}}
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
