#!/usr/bin/env python3
"""How does the seeding batch scale when K host threads run it side by side, each on its own context and stream -- the
preload shim's phase 1?  Aggregate reads/s of bmh_smem_batch (host buffers in and out, min_emit_len = min_seed_len) for
K = 1..16 threads over batches of --batch reads.  Needs oracle/_ref (builds the index).
Usage (GPU box): python tools/smem_concurrency.py [--genome 4600000] [--batch 33334] [--rounds 6]"""
import argparse
import ctypes as C
import os
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kswgen  # noqa: E402
import kswlib  # noqa: E402
import reflib  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--genome", type=int, default=4_600_000)
ap.add_argument("--batch", type=int, default=33334)
ap.add_argument("--rounds", type=int, default=6)
a = ap.parse_args()
pkg = load_package()
lib = pkg.lib()
rng = np.random.default_rng(5)
tmp = tempfile.mkdtemp(prefix="bmh_smc_")
ref = kswgen.rand_seq(rng, a.genome)
fa = os.path.join(tmp, "ref.fa")
reflib.write_fasta(fa, "synth", ref)
reflib.build_index(fa)
idx = reflib.lib().bwa_idx_load(fa.encode(), 7)
prim, L2, sl, words, sai, sa = reflib.bwt_arrays(idx)
so = np.zeros((), dtype=pkg.SMEM_OPT)
o = reflib.smem_opt_of(reflib.opt_from_params(kswlib.make_params()))
for k in o.dtype.names:
    so[k] = o[k]
so["min_emit_len"] = so["min_seed_len"]
L, n = 150, a.batch
pos = rng.integers(0, a.genome - L - 8, size=n)
reads = ref[pos[:, None] + np.arange(L)[None, :]]
rate = np.where(rng.random(n) < 0.15, 0.12, 0.02)[:, None]
sub = rng.random(reads.shape) < rate
reads = np.ascontiguousarray(np.where(sub, (reads + rng.integers(1, 4, reads.shape)) & 3, reads).astype(np.uint8))


class Read(C.Structure):
    _fields_ = [("l_seq", C.c_int), ("seq", C.c_void_p)]


c_reads = (Read * n)()
for k in range(n):
    c_reads[k].l_seq, c_reads[k].seq = L, reads[k].ctypes.data
KMAX = 16
ctxs = [pkg.Context(0, kswlib.make_params()) for _ in range(KMAX)]
for c in ctxs:
    c.set_bwt(prim, L2, sl, words, sai, sa)
bufs = []
for _ in range(KMAX):
    bufs.append((np.zeros(n + 1, np.uint32), np.zeros(n * 16, pkg.SMEM_CALL), np.zeros(n + 1, np.uint64), np.zeros(n * 8, pkg.SMEM_INTV)))


def work(k, rounds):
    co, ca, io, iv = bufs[k]
    for _ in range(rounds):
        rc = lib.bmh_smem_batch(ctxs[k]._h, so.ctypes.data_as(C.c_void_p), n, C.cast(c_reads, C.c_void_p), co.ctypes.data_as(C.c_void_p),
                                ca.ctypes.data_as(C.c_void_p), C.c_size_t(len(ca)), io.ctypes.data_as(C.c_void_p), iv.ctypes.data_as(C.c_void_p),
                                C.c_size_t(len(iv)))
        assert rc == 0, rc


work(0, 2)
for K in (1, 2, 4, 8, 12, 16):
    for k in range(K):
        work(k, 1)
    th = [threading.Thread(target=work, args=(k, a.rounds)) for k in range(K)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    print(f"{K:2d} threads x {a.rounds} batches of {n} reads: {dt * 1e3 / a.rounds:7.2f} ms per round, {K * a.rounds * n / dt / 1e6:6.2f} M reads/s", flush=True)
