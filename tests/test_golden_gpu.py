"""The HIP path (through the C-ABI) against the committed golden fixtures produced by the compiled
reference.  Bit-exact or fail."""
import numpy as np
import pytest

import kswlib
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    pkg = load_package()
    c = pkg.Context(0, kswlib.make_params())
    yield c
    c.close()


def test_hip_extend_matches_reference_fixture(ctx):
    g = kswlib.load_golden("ext_golden.npz")
    pool, tasks, exp, grp, params = g["pool"], g["tasks"], g["expect"], g["group"], g["params"]
    for k in range(len(params)):
        sel = np.nonzero(grp == k)[0]
        ctx.set_params(params[k])
        got = ctx.extend_batch(pool, tasks[sel])
        bad = np.nonzero(got != exp[sel])[0]
        assert len(bad) == 0, f"set {k} task {tasks[sel][bad[0]]}: gpu={got[bad[0]]} ref={exp[sel][bad[0]]}"


def test_hip_global_matches_reference_fixture(ctx):
    g = kswlib.load_golden("glb_golden.npz")
    pool, tasks, exp, grp, params, cigar = g["pool"], g["tasks"], g["expect"], g["group"], g["params"], g["cigar"]
    for k in range(len(params)):
        sel = np.nonzero(grp == k)[0]
        ctx.set_params(params[k])
        res, cig = ctx.global_batch(pool, tasks[sel], len(cigar))
        assert (res == exp[sel]).all()
        for t, r in zip(tasks[sel], res):
            o, n = int(t["cigar_off"]), int(r["n_cigar"])
            assert np.array_equal(cig[o:o + n], cigar[o:o + n])


def test_hip_chain2aln_driver_matches_reference_fixture(ctx):
    """bmh_chain2aln_batch (batched, GPU rounds) == the reference's sequential mem_chain2aln."""
    nreg = 0
    for p, l_pac, pac, reads, chains, exp in kswlib.golden_chain2aln_groups():
        ctx.set_params(p)
        got = ctx.chain2aln_batch(l_pac, pac, reads, chains)
        for r, (a, b) in enumerate(zip(got, exp)):
            assert len(a) == len(b) and (a == b).all(), f"read {r}: gpu={a} ref={b}"
            nreg += len(b)
        st = ctx.driver_stats()
        assert st["rounds"] >= 2 and st["ext_tasks"] > 0
    assert nreg >= 2500


def test_hip_reg2cigar_driver_matches_reference_fixture(ctx):
    """bmh_reg2cigar_batch (batched GPU rounds) == the reference's mem_reg2aln (CIGAR, NM, MD), and == the oracle
    on score and number of tries."""
    n = 0
    for p, l_pac, pac, reads, reqs, exp in kswlib.golden_cigar_groups():
        ctx.set_params(p)
        res, cig, md = ctx.reg2cigar_batch(l_pac, pac, reads, reqs)
        mdb = bytes(md)
        for rq, r, (en, ew, enm, emd) in zip(reqs, res, exp):
            words = cig[int(r["cigar_off"]): int(r["cigar_off"]) + int(r["n_cigar"])]
            m = mdb[int(r["md_off"]): int(r["md_off"]) + int(r["md_len"])]
            read = reads[int(rq["read"])]
            fw, fmd = kswlib.finish_aln(words, m, rq, len(read), l_pac)
            assert len(fw) == en and np.array_equal(fw, ew), f"req {rq}: gpu={fw} ref={ew}"
            assert int(r["NM"]) == enm and fmd == emd
            if n % 50 == 0:
                oscore, _, _, _, otries = kswlib.orc_reg2cigar(p, l_pac, pac, read, rq)
                assert int(r["score"]) == oscore and int(r["tries"]) == otries
            n += 1
    assert n >= 2000
