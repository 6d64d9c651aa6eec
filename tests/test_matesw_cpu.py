"""Mate rescue (the block of mem_sam_pe at reference bwamem_pair.c:251-263 over mem_matesw :109-175): the CPU
restatement oracle/matesw_oracle.c against the reference's own functions and against the committed fixture.
mem_sort_and_dedup is the reference's own function throughout (it lies outside the path), hence the gate."""
import numpy as np
import pytest

import kswlib
import reflib

pytestmark = pytest.mark.skipif(not reflib.have_ref_bwa(), reason="oracle/_ref not built (no /root/reference here)")


def _same(got, exp):
    return all(len(a) == len(b) and (a == b).all() for a, b in zip(got, exp))


def test_oracle_matesw_matches_reference_fixture():
    n_calls = 0
    for p, o, pes, l_pac, pac, reads, regs, exp, n_sw in kswlib.golden_matesw_groups():
        opt = reflib.opt_from_params(p)
        got, ns = kswlib.orc_matesw_pairs(p, o, l_pac, pac, pes, reads, regs, reflib.ref_dedup_fn(opt))
        assert ns == n_sw
        assert _same(got, exp)
        n_calls += sum(ns)
    assert n_calls > 4000


def test_oracle_matesw_matches_reference_live(tmp_path):
    import importlib
    import os
    import sys
    sys.path.insert(0, os.path.join(kswlib.ROOT, "tools"))
    mk = importlib.import_module("make_matesw_fixture")
    import kswgen
    rng = np.random.default_rng(171)
    ref = kswgen.rand_seq(rng, 120000)
    fa = str(tmp_path / "ref.fa")
    reflib.write_fasta(fa, "synth", ref)
    reflib.build_index(fa)
    idx = reflib.lib().bwa_idx_load(fa.encode(), 7)
    l_pac, pac = reflib.pac_of(idx)
    p = kswlib.make_params()
    opt = reflib.opt_from_params(p)
    reads = mk.sim_pairs(rng, ref, 150, 130)
    regs = reflib.ref_align_reads(idx, opt, reads)
    pes = reflib.ref_pestat(idx, opt, regs)
    pes["failed"] = 0
    pes["low"], pes["high"] = pes["low"][1], pes["high"][1]
    exp, ns = reflib.ref_matesw_pairs(idx, opt, pes, reads, regs)
    o = np.zeros((), kswlib.MATESW_OPT)
    o["pen_unpaired"], o["max_matesw"], o["min_seed_len"] = opt.contents.pen_unpaired, opt.contents.max_matesw, opt.contents.min_seed_len
    got, ns2 = kswlib.orc_matesw_pairs(p, o, l_pac, pac, pes, reads, regs, reflib.ref_dedup_fn(opt))
    assert ns == ns2 and sum(ns) > 500
    assert _same(got, exp)
