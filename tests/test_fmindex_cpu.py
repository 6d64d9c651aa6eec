"""FM-index queries of the seeding stage: the CPU restatement oracle/fmindex_oracle.c against
the committed fixture (outputs of the reference's bwt_smem1 / bwt_sa on an index the reference built) and, where the
compiled reference is present, against the reference's functions live -- including smem_next2's merged output, which
pins the ORDER of bwt_smem1 calls the restatement makes."""
import numpy as np
import pytest

import kswlib
import reflib


def _same_calls(got, want):
    (gc, gi), (wc, wi) = got, want
    return len(gc) == len(wc) and all((gc[f] == wc[f]).all() for f in ("x", "min_intv", "ret", "n", "first")) and \
        len(gi) == len(wi) and (gi == wi).all()


def test_oracle_fmindex_matches_reference_fixture():
    cb, keep, raw, opt, reads, per, sa_k, sa_pos = kswlib.golden_fmindex()
    assert len(reads) >= 400 and sum(len(c) for c, _ in per) > 4000
    for rd, want in zip(reads, per):
        assert _same_calls(kswlib.orc_smem_calls(cb, opt, rd), want)
    assert (kswlib.orc_sa(cb, sa_k) == sa_pos).all()


@pytest.mark.skipif(not reflib.have_ref_bwa(), reason="oracle/_ref not built (no /root/reference here)")
def test_oracle_fmindex_matches_reference_live(tmp_path):
    import kswgen
    rng = np.random.default_rng(181)
    ref = kswgen.rand_seq(rng, 50000)
    for _ in range(15):
        a, b, L = int(rng.integers(0, 47000)), int(rng.integers(0, 47000)), int(rng.integers(80, 400))
        ref[b:b + L] = kswgen.mutate(rng, ref[a:a + L + 20], 0.01, 0.002, 0.002, 2)[:L]
    fa = str(tmp_path / "ref.fa")
    reflib.write_fasta(fa, "synth", ref)
    reflib.build_index(fa)
    idx = reflib.lib().bwa_idx_load(fa.encode(), 7)
    prim, L2, sl, words, sai, sa = reflib.bwt_arrays(idx)
    keep = []
    cb = kswlib.make_cbwt(prim, L2, sl, words, sai, sa, keep)
    opt = reflib.opt_from_params(kswlib.make_params())
    so = reflib.smem_opt_of(opt)
    ks = rng.integers(0, sl + 1, 2000).astype(np.uint64)
    assert (kswlib.orc_sa(cb, ks) == reflib.ref_sa(idx, ks)).all()
    n_calls = 0
    for it in range(150):
        Lr = int(rng.integers(30, 250))
        pos = int(rng.integers(0, len(ref) - Lr - 12))
        rd = kswgen.mutate(rng, ref[pos:pos + Lr + 10], 0.03, 0.004, 0.004, 3)[:Lr].copy()
        if it % 3 == 0:
            rd[rng.random(len(rd)) < 0.03] = 4
        calls, pool = kswlib.orc_smem_calls(cb, so, rd)
        n_calls += len(calls)
        for c in calls:  # every call == the reference's bwt_smem1 with the same arguments
            ret, iv = reflib.ref_smem1(idx, rd, int(c["x"]), int(c["min_intv"]))
            mine = pool[int(c["first"]): int(c["first"]) + int(c["n"])]
            assert ret == int(c["ret"]) and len(iv) == len(mine) and (iv == mine).all()
        # ... and the calls are the ones smem_next2 makes: every interval it returns is one of ours, every main call's
        # return value is where its next iteration starts
        its = reflib.ref_smem_iter(idx, opt, rd)
        mains = [c for c in calls if int(c["min_intv"]) == int(so["start_width"])]
        assert len(mains) == len(its)
        have = {tuple(int(v[f]) for f in ("x0", "x1", "x2", "info")) for v in pool}
        for v in its:
            assert all(tuple(int(w[f]) for f in ("x0", "x1", "x2", "info")) in have for w in v)
    assert n_calls > 1000


def test_oracle_min_emit_len_is_a_pure_filter():
    """bmh_smem_opt_t.min_emit_len hands back the long intervals of the same calls, in the same order."""
    cb, keep, raw, opt, reads, per, sa_k, sa_pos = kswlib.golden_fmindex()
    o2 = np.array(opt, dtype=kswlib.SMEM_OPT).copy()
    o2["min_emit_len"] = int(o2["min_seed_len"])
    n_short = 0
    for rd in reads:
        calls, pool = kswlib.orc_smem_calls(cb, opt, rd)
        fc, fp = kswlib.orc_smem_calls(cb, o2, rd)
        assert len(fc) == len(calls)
        want = []
        for c, f in zip(calls, fc):
            assert all(int(c[k]) == int(f[k]) for k in ("x", "min_intv", "ret"))
            iv = pool[int(c["first"]): int(c["first"]) + int(c["n"])]
            ln = (iv["info"] & np.uint64(0xffffffff)).astype(np.int64) - (iv["info"] >> np.uint64(32)).astype(np.int64)
            long_ = iv[ln >= int(o2["min_emit_len"])]
            n_short += len(iv) - len(long_)
            assert int(f["n"]) == len(long_) and (fp[int(f["first"]): int(f["first"]) + int(f["n"])] == long_).all()
            want.append(long_)
        assert len(fp) == sum(len(w) for w in want)
    assert n_short > 100
