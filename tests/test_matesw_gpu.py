"""Batched mate rescue on the GPU: bmh_matesw_batch == the reference's own per-pair loop (bwamem_pair.c:251-263 over
mem_matesw), with the window read from the resident 2-bit reference or decoded on the host."""
import numpy as np
import pytest

import kswlib
import reflib
from test_kernel_families_gpu import _ctx_with

pytestmark = pytest.mark.gpu


def _same(got, exp):
    for k, (a, b) in enumerate(zip(got, exp)):
        assert len(a) == len(b) and (a == b).all(), f"vector {k} (pair {k // 2}): gpu={a} want={b}"


@pytest.mark.skipif(not reflib.have_ref_bwa(), reason="oracle/_ref not built: no mem_sort_and_dedup to pass in")
@pytest.mark.parametrize("resident", [True, False])
def test_matesw_batch_matches_reference_fixture(resident):
    ctx = _ctx_with({})
    calls = 0
    for p, o, pes, l_pac, pac, reads, regs, exp, n_sw in kswlib.golden_matesw_groups():
        ctx.set_params(p)
        if resident:
            pac = ctx.set_pac(pac, l_pac)
        opt = reflib.opt_from_params(p)
        got, ns = ctx.matesw_batch(l_pac, pac, reads, regs, pes, o, reflib.ref_dedup_fn(opt))
        assert ns == n_sw
        _same(got, exp)
        calls += sum(ns)
    assert calls > 4000
    ctx.close()


def test_matesw_batch_matches_oracle_without_the_reference():
    """Same inputs, but mem_sort_and_dedup replaced on BOTH sides by the oracle's deterministic stand-in, so the
    comparison needs nothing from the reference build."""
    ctx = _ctx_with({})
    dd = kswlib.simple_dedup_fn()
    for p, o, pes, l_pac, pac, reads, regs, _exp, _n in kswlib.golden_matesw_groups():
        ctx.set_params(p)
        pac = ctx.set_pac(pac, l_pac)
        want, wn = kswlib.orc_matesw_pairs(p, o, l_pac, pac, pes, reads, regs, dd)
        got, gn = ctx.matesw_batch(l_pac, pac, reads, regs, pes, o, dd)
        assert gn == wn
        _same(got, want)
    ctx.close()
