"""The CPU oracle against the committed golden fixtures (tests/golden/*.npz), which were produced by
the compiled REFERENCE (tools/make_golden.py).  This is what pins the oracle on machines where the
reference itself is not available (the GPU box)."""
import numpy as np

import kswlib


def test_oracle_extend_matches_reference_fixture():
    g = kswlib.load_golden("ext_golden.npz")
    pool, tasks, exp, grp, params = g["pool"], g["tasks"], g["expect"], g["group"], g["params"]
    assert len(tasks) >= 5000
    for k in range(len(params)):
        sel = np.nonzero(grp == k)[0]
        got, _ = kswlib.orc_extend_batch(params[k], pool, tasks[sel])
        assert (got == exp[sel]).all(), f"parameter set {k}"


def test_oracle_global_matches_reference_fixture():
    g = kswlib.load_golden("glb_golden.npz")
    pool, tasks, exp, grp, params, cigar = g["pool"], g["tasks"], g["expect"], g["group"], g["params"], g["cigar"]
    assert len(tasks) >= 2000
    for k in range(len(params)):
        sel = np.nonzero(grp == k)[0]
        res, cigs = kswlib.orc_global_batch(params[k], pool, tasks[sel])
        assert (res == exp[sel]).all()
        for t, r, c in zip(tasks[sel], res, cigs):
            o = int(t["cigar_off"])
            assert np.array_equal(cigar[o:o + int(r["n_cigar"])], c)


def test_oracle_chain2aln_matches_reference_fixture():
    n = 0
    for p, l_pac, pac, reads, chains, exp in kswlib.golden_chain2aln_groups():
        got = kswlib.orc_chain2aln_reads(p, l_pac, pac, reads, chains)
        for r, (a, b) in enumerate(zip(got, exp)):
            assert len(a) == len(b) and (a == b).all(), f"read {r}: {a} vs {b}"
            n += len(b)
    assert n >= 2500


def test_fixture_coverage_matrix():
    """The fixtures hit the cases SURVEY §8c lists (band retries, z-drop exits, tiny and empty inputs ...)."""
    g = kswlib.load_golden("ext_golden.npz")
    t, e, p = g["tasks"], g["expect"], g["params"][g["group"]]
    assert (t["qlen"] == 1).any() and (t["tlen"] == 0).any() and (t["tlen"] <= 2).sum() > 10
    assert (t["h0"] == 0).any() and (t["qlen"] > 128).any() and (t["tlen"] > 256).any()
    assert (e["max_off"] >= (t["w"] >> 1) + (t["w"] >> 2)).sum() > 50      # would trigger the 2w retry (bwamem.c:828)
    assert (p["zdrop"] <= 0).any() and (p["e_ins"] != p["e_del"]).any() and (p["a"] == 2).any()
    assert (e["gscore"] == -1).any() and (e["gscore"] > 0).any() and (e["tle"] == 0).any()
    assert ((t["flags"] & 1) > 0).any() and ((t["flags"] & 2) > 0).any()
    assert (t["tlen"].astype(int) > t["qlen"].astype(int) + t["w"].astype(int)).sum() > 100  # empty-row territory


def test_oracle_reg2cigar_matches_reference_mem_reg2aln_fixture():
    """orc_reg2cigar (band inference + retry loop + bwa_gen_cigar2 restatement) followed by the reference's
    post-processing (clipping, terminal deletions) == the reference's own mem_reg2aln."""
    n = 0
    for p, l_pac, pac, reads, reqs, exp in kswlib.golden_cigar_groups():
        for rq, (en, ew, enm, emd) in zip(reqs, exp):
            read = reads[int(rq["read"])]
            score, words, nm, md, tries = kswlib.orc_reg2cigar(p, l_pac, pac, read, rq)
            fw, fmd = kswlib.finish_aln(words, md, rq, len(read), l_pac)
            assert len(fw) == en and np.array_equal(fw, ew), f"req {rq}: {fw} vs {ew}"
            assert nm == enm and fmd == emd
            n += 1
    assert n >= 2000


def test_oracle_sw_matches_reference_fixture():
    """orc_align2 == the reference's ksw_align2 (ksw.c:341-364) on the committed mate-rescue / fuzz vectors."""
    g = kswlib.load_golden("sw_golden.npz")
    pool, tasks, exp, grp, params = g["pool"], g["tasks"], g["expect"], g["group"], g["params"]
    assert len(tasks) >= 2000 and (exp["score2"] > 0).sum() > 200 and (exp["tb"] >= 0).sum() > 500
    for k in range(len(params)):
        sel = np.nonzero(grp == k)[0]
        got, _ = kswlib.orc_sw_batch(params[k], pool, tasks[sel], nthreads=4)
        assert (got["rsv"] == 0).all()
        for f in kswlib.SW_FIELDS:
            assert (got[f] == exp[sel][f]).all(), f"parameter set {k}, field {f}"
