#!/usr/bin/env python3
"""Generate tests/golden/matesw_golden.npz FROM THE REFERENCE ITSELF (build container only).

Inputs : read pairs simulated from a synthetic genome (half of the second mates too noisy to be seeded), their
         regions from the reference's own phase 1 (mem_align1_core, bwamem.c:1122) and two insert-size tables:
         the one the reference infers (mem_pestat, bwamem_pair.c:46) and one with all four orientations open.
Outputs: the region vectors after the reference's own mate rescue (the block of mem_sam_pe at bwamem_pair.c:251-263,
         driven with the reference's mem_matesw and mem_sort_and_dedup) and its per-pair SW counts.
The fixture is data; no reference source is stored.  Usage: python tools/make_matesw_fixture.py"""
import importlib
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import kswgen  # noqa: E402
import kswlib  # noqa: E402
import reflib  # noqa: E402


def sim_pairs(rng, ref, n, L):
    reads = []
    for _ in range(n):
        ins = int(rng.integers(250, 450))
        pos = int(rng.integers(0, len(ref) - ins - 60))
        frag = ref[pos:pos + ins + 40]
        a = kswgen.mutate(rng, frag[:L + 30], 0.02, 0.0025, 0.0025, 1)[:L]
        noisy = rng.random() < 0.5
        b = kswgen.mutate(rng, frag[ins - L:ins + 30], 0.16 if noisy else 0.02, 0.004, 0.004, 2)[:L]
        b = (3 - b[::-1]).astype(np.uint8)
        if rng.random() < 0.03:
            a[rng.random(len(a)) < 0.03] = 4
        if rng.random() < 0.5:  # either mate may be the anchor
            a, b = b, a
        reads += [a.copy(), b.copy()]
    return reads


def pack_regs(regs):
    return np.concatenate(regs) if sum(len(r) for r in regs) else np.zeros(0, kswlib.ALNREG), np.array([len(r) for r in regs], np.int32)


def main():
    assert reflib.have_ref_bwa()
    rng = np.random.default_rng(20261008)
    tmp = tempfile.mkdtemp(prefix="bmh_msw_")
    ref = kswgen.rand_seq(rng, 250000)
    fa = os.path.join(tmp, "ref.fa")
    reflib.write_fasta(fa, "synth", ref)
    reflib.build_index(fa)
    idx = reflib.lib().bwa_idx_load(fa.encode(), 7)
    l_pac, pac = reflib.pac_of(idx)
    out = {"l_pac": l_pac, "pac": pac}
    groups = []
    for g, (p, L) in enumerate([(kswlib.make_params(), 150), (kswlib.make_params(), 100), (kswlib.make_params(a=2, b=5, o_del=8, o_ins=8), 120)]):
        opt = reflib.opt_from_params(p)
        opt.contents.b = 5 if g == 2 else 4
        reads = sim_pairs(rng, ref, 260, L)
        regs = reflib.ref_align_reads(idx, opt, reads)
        pes_ref = reflib.ref_pestat(idx, opt, regs)
        pes_open = pes_ref.copy()
        pes_open["failed"] = 0
        pes_open["low"], pes_open["high"] = pes_ref["low"][1], pes_ref["high"][1]
        for v, pes in enumerate((pes_ref, pes_open)):
            exp, ns = reflib.ref_matesw_pairs(idx, opt, pes, reads, regs)
            o = np.zeros((), kswlib.MATESW_OPT)
            o["pen_unpaired"], o["max_matesw"], o["min_seed_len"] = opt.contents.pen_unpaired, opt.contents.max_matesw, opt.contents.min_seed_len
            key = f"g{g}v{v}_"
            out[key + "params"], out[key + "opt"], out[key + "pes"] = np.array(p), o, pes
            out[key + "reads"] = np.concatenate(reads)
            out[key + "read_len"] = np.array([len(r) for r in reads], np.int32)
            out[key + "regs"], out[key + "regs_n"] = pack_regs(regs)
            out[key + "exp"], out[key + "exp_n"] = pack_regs(exp)
            out[key + "n_sw"] = np.array(ns, np.int32)
            out[key + "mask_level_redun"] = np.float32(opt.contents.mask_level_redun)
            groups.append(key)
            print(key, "pairs", len(reads) // 2, "SW calls", sum(ns), "regions", sum(len(r) for r in regs), "->", sum(len(r) for r in exp))
    out["groups"] = np.array(groups)
    np.savez_compressed(os.path.join(kswlib.GOLDEN_DIR, "matesw_golden.npz"), **out)
    print(os.path.getsize(os.path.join(kswlib.GOLDEN_DIR, "matesw_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
