#!/usr/bin/env python3
"""Fixture for seeds-to-chains (bmh_chain_reads): a repeat-rich synthetic genome indexed by the
COMPILED REFERENCE (`oracle/_ref/bwa index`), reads drawn from unique sequence and from repeat families (so that a read
has from one to dozens of chains: B-tree splits, equal keys, equal weights), and for every read
  inputs    the bwt_smem1 call records + intervals of smem_next2's iteration (what bmh_smem_batch returns; produced
            here by the CPU oracle, which tests/test_fmindex_cpu.py pins against the reference) and the bwt_sa table
  expected  the chains of the reference's own mem_chain + mem_chain_flt (bwamem.c:283-380), in its order.
Output: tests/golden/chain_golden.npz (data only).  Run in the build container: python tools/make_chain_fixture.py"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kswgen  # noqa: E402
import kswlib  # noqa: E402
import reflib  # noqa: E402


def build(rng, G=600_000):
    ref = kswgen.rand_seq(rng, G)
    fams = []
    for _ in range(12):  # repeat families: 6-60 copies of a 150-900 bp element, 0-4 % diverged
        L, copies = int(rng.integers(150, 900)), int(rng.choice([6, 12, 25, 60]))
        src = kswgen.rand_seq(rng, L + 60)
        locs = []
        for _c in range(copies):
            dst = int(rng.integers(1000, G - 2000))
            ref[dst:dst + L] = kswgen.mutate(rng, src, float(rng.choice([0.0, 0.01, 0.04])), 0.001, 0.001, 2)[:L]
            locs.append((dst, L))
        fams.append(locs)
    return ref, fams


def main():
    rng = np.random.default_rng(20261201)
    tmp = tempfile.mkdtemp(prefix="bmh_chain_")
    ref, fams = build(rng)
    fa = os.path.join(tmp, "ref.fa")
    reflib.write_fasta(fa, "synth", ref)
    reflib.build_index(fa)
    L = reflib.lib()
    idx = L.bwa_idx_load(fa.encode(), 7)
    l_pac = int(idx.contents.bns.contents.l_pac)
    prim, L2, sl, words, sai, sa = reflib.bwt_arrays(idx)
    keep = []
    cb = kswlib.make_cbwt(prim, L2, sl, words, sai, sa, keep)
    out = {"l_pac": np.int64(l_pac)}
    groups = []
    for gi, kw in enumerate([dict(), dict(min_seed_len=15, max_occ=40), dict(w=20, max_chain_gap=200, chain_drop_ratio=0.3, mask_level=0.8)]):
        opt = L.mem_opt_init()
        for k, v in kw.items():
            setattr(opt.contents, k, v)
        so = reflib.smem_opt_of(opt)
        reads = []
        for _ in range(260):
            Lr = int(rng.choice([100, 150, 150, 250]))
            if rng.random() < 0.6:  # from a repeat copy (possibly hanging over its edge)
                dst, Lf = fams[int(rng.integers(0, len(fams)))][0]
                pos = dst + int(rng.integers(-Lr // 2, max(1, Lf - Lr // 2)))
            else:
                pos = int(rng.integers(0, len(ref) - Lr - 8))
            pos = min(max(pos, 0), len(ref) - Lr - 8)
            r = kswgen.mutate(rng, ref[pos:pos + Lr + 20], 0.02, 0.003, 0.003, 3)[:Lr]
            if rng.random() < 0.5:
                r = (3 - r[::-1]).astype(np.uint8)
            if rng.random() < 0.1:
                r[int(rng.integers(0, Lr))] = 4
            reads.append(np.ascontiguousarray(r, dtype=np.uint8))
        calls, intvs, ks = [], [], []
        for r in reads:
            c, iv = kswlib.orc_smem_calls(cb, so, r)
            calls.append(c)
            intvs.append(iv)
            ln = (iv["info"] & 0xffffffff).astype(np.int64) - (iv["info"] >> 32).astype(np.int64)
            sel = (ln >= opt.contents.min_seed_len) & (iv["x2"] <= opt.contents.max_occ)
            for x0, x2 in zip(iv["x0"][sel], iv["x2"][sel]):
                ks.append(np.arange(int(x0), int(x0) + int(x2), dtype=np.uint64))
        ks = np.unique(np.concatenate(ks)) if ks else np.zeros(0, np.uint64)
        pos = reflib.ref_sa(idx, ks)
        chains, _ = reflib.chains_and_regs(idx, opt, reads, run_chain2aln=False)
        nch = np.array([len(c) for c in chains], dtype=np.int32)
        nseed = np.array([len(s) for c in chains for s in c], dtype=np.int32)
        seeds = np.concatenate([s for c in chains for s in c]) if nseed.sum() else np.zeros(0, kswlib.SEED)
        p = f"g{gi}_"
        groups.append(p)
        o = opt.contents
        out[p + "opt"] = np.array([o.w, o.max_chain_gap, o.min_seed_len, o.max_occ, int(so["split_len"]), o.split_width], dtype=np.int32)
        out[p + "optf"] = np.array([o.mask_level, o.chain_drop_ratio], dtype=np.float32)
        out[p + "reads"], out[p + "read_len"] = np.concatenate(reads), np.array([len(r) for r in reads], dtype=np.int32)
        out[p + "calls"], out[p + "n_calls"] = np.concatenate(calls), np.array([len(c) for c in calls], dtype=np.int32)
        out[p + "intv"], out[p + "n_intv"] = np.concatenate(intvs), np.array([len(v) for v in intvs], dtype=np.int64)
        out[p + "sa_k"], out[p + "sa_pos"] = ks, pos
        out[p + "n_chains"], out[p + "n_seeds"], out[p + "seeds"] = nch, nseed, seeds
        print(p, "reads", len(reads), "chains", int(nch.sum()), "max chains/read", int(nch.max()), "seeds", int(nseed.sum()), "sa", len(ks))
    out["groups"] = np.array(groups)
    path = os.path.join(ROOT, "tests", "golden", "chain_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
