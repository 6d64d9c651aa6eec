"""The HIP path (through the C-ABI) against the committed golden fixtures produced by the compiled
reference.  Bit-exact or fail."""
import numpy as np
import pytest

import kswlib
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["default", "masked"])
def ctx(request):
    """"masked": the global lane kernels without the unmasked body for blocks inside every lane's band."""
    from test_kernel_families_gpu import _ctx_with
    c = _ctx_with({} if request.param == "default" else {"BMH_GL_FAST": "0"})
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
        # one fused round for the first seed of every chain, at most one more for whatever a read still needs
        # (SURVEY.md §8 row a5); the extensions run are a superset of the reference's
        assert 1 <= st["rounds"] <= 2 and st["ext_tasks"] > 0, st
        assert st["seeds_extended"] == sum(len(b) for b in exp) and st["seeds_speculated"] >= 0, st
    assert nreg >= 2500


@pytest.mark.parametrize("path", ["host copies", "region records"])
def test_hip_reg2cigar_driver_matches_reference_fixture(ctx, path):
    """bmh_reg2cigar_batch (batched GPU rounds) == the reference's mem_reg2aln (CIGAR, NM, MD), and == the oracle
    on score and number of tries.  "region records": with the 2-bit reference resident on the device the driver hands window fetch,
    orientation, the no-gap score, the replay of the band loop, NM and MD to bmh_region_cigar_batch."""
    n = 0
    for p, l_pac, pac, reads, reqs, exp in kswlib.golden_cigar_groups():
        ctx.set_params(p)
        if path == "region records":
            pac = ctx.set_pac(pac, l_pac)
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


@pytest.mark.parametrize("path", ["host copies", "region records"])
def test_hip_reg2cigar_long_cigars_take_the_retry_path(ctx, path):
    """The driver reserves few CIGAR slots per task and redoes a task whose CIGAR needs more: reads with an indel every
    ~25 bases (50+ operations), both strands, mixed with ordinary ones, against the oracle.  With region records the long ones (and
    those whose MD outgrows its slot) come back flagged and are redone through the path with host copies."""
    rng = np.random.default_rng(201)
    l_pac = 20000
    bases = rng.integers(0, 4, l_pac, dtype=np.uint8)
    pad = np.concatenate([bases, np.zeros(4, np.uint8)])
    q4 = pad[: (len(pad) // 4) * 4].reshape(-1, 4)
    pac = (q4[:, 0] << 6 | q4[:, 1] << 4 | q4[:, 2] << 2 | q4[:, 3]).astype(np.uint8)
    p = kswlib.make_params()
    ctx.set_params(p)
    reads, reqs = [], []
    for k in range(60):
        L = int(rng.integers(300, 640))
        pos = int(rng.integers(0, l_pac - L - 50))
        src = bases[pos:pos + L + 40]
        out, i, since = [], 0, 0
        step = 25 if k % 3 else 400  # every third read is an ordinary one
        while len(out) < L and i < len(src):
            since += 1
            if since >= step:
                since = 0
                if rng.random() < 0.5:
                    i += 1  # deletion from the read
                else:
                    out.append(int(rng.integers(0, 4)))  # insertion
                    continue
            out.append(int(src[i]))
            i += 1
        rd = np.array(out[:L], dtype=np.uint8)
        rb, re = pos, pos + i
        rq = np.zeros((), kswlib.CIGAR_REQ)
        if k % 2:  # reverse strand: the read is the reverse complement, the region lies in [l_pac, 2*l_pac)
            rd = (3 - rd[::-1]).astype(np.uint8)
            rb, re = 2 * l_pac - re, 2 * l_pac - rb
        rq["read"], rq["qb"], rq["qe"], rq["rb"], rq["re"], rq["truesc"], rq["reg_w"] = len(reads), 0, len(rd), rb, re, len(rd) - 60, 100
        reads.append(rd), reqs.append(rq)
    reqs = np.array(reqs)
    if path == "region records":
        pac = ctx.set_pac(pac, l_pac)
    res, cig, md = ctx.reg2cigar_batch(l_pac, pac, reads, reqs)
    mdb = bytes(md)
    long_ones = 0
    for rq, r in zip(reqs, res):
        oscore, owords, onm, omd, otries = kswlib.orc_reg2cigar(p, l_pac, pac, reads[int(rq["read"])], rq)
        words = cig[int(r["cigar_off"]): int(r["cigar_off"]) + int(r["n_cigar"])]
        assert int(r["score"]) == oscore and int(r["tries"]) == otries and int(r["NM"]) == onm
        assert np.array_equal(words, owords)
        assert mdb[int(r["md_off"]): int(r["md_off"]) + int(r["md_len"])] == omd.rstrip(b"\0")
        long_ones += len(owords) > 24
    assert long_ones >= 30
