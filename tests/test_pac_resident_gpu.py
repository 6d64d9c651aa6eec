"""SURVEY.md §8(f) row 1 -- bns_get_seq on the device: extension tasks flagged BMH_F_TPAC take their target straight
from the 2-bit reference resident in HBM (doubled coordinate, reference bntseq.c:355-376) instead of pool bytes."""
import numpy as np
import pytest

import kswlib
from kswlib import BMH_F_QREV, BMH_F_TREV, BMH_F_TPAC, EXT_TASK
from test_kernel_families_gpu import _ctx_with

pytestmark = pytest.mark.gpu


def _pac_tasks(rng, n, l_pac, read_len):
    """Random reference + reads copied from either strand with errors; one left and one right extension per read
    addressed (a) by doubled-coordinate position with BMH_F_TPAC and (b) by window bytes in the pool."""
    bases = rng.integers(0, 4, l_pac, dtype=np.uint8)
    pad = np.concatenate([bases, np.zeros((-l_pac) % 4 + 4, np.uint8)])
    q = pad[: (len(pad) // 4) * 4].reshape(-1, 4)
    pac = (q[:, 0] << 6 | q[:, 1] << 4 | q[:, 2] << 2 | q[:, 3]).astype(np.uint8)
    dbl = np.concatenate([bases, (3 - bases)[::-1]])  # the doubled coordinate, one code per byte
    pool, tp, tb = [], [], []
    off = 0
    for _ in range(n):
        L = int(rng.integers(read_len[0], read_len[1] + 1))
        rev = int(rng.integers(0, 2))
        lo, hi = (l_pac, 2 * l_pac) if rev else (0, l_pac)
        pos = int(rng.integers(lo + 200, hi - L - 200))
        read = dbl[pos:pos + L].copy()
        mut = rng.random(L) < 0.04
        read[mut] = (read[mut] + rng.integers(1, 4, mut.sum())) & 3
        if rng.random() < 0.3:  # a small deletion in the read
            c = int(rng.integers(5, L - 5))
            read = np.concatenate([read[:c], read[c + int(rng.integers(1, 4)):]])
            L = len(read)
        qb = int(rng.integers(1, L - 25))
        ql = int(rng.integers(19, L - qb)) if L - qb > 19 else L - qb
        rb = pos + qb
        ext = 100 + L
        w0 = max(lo, rb - qb - ext)
        w1 = min(hi, rb + ql + (L - qb - ql) + ext)
        if rng.random() < 0.1:  # windows clipped by the strand boundary, bwamem.c:749-755
            w0 = max(w0, rb - int(rng.integers(0, qb + 1)))
        win = dbl[w0:w1]
        read_off, win_off = off, off + L
        pool += [read, win]
        off += L + len(win)
        for left in (True, False):
            a, b = np.zeros((), EXT_TASK), np.zeros((), EXT_TASK)
            if left:
                tl = rb - w0
                a["q_off"] = b["q_off"] = read_off + qb - 1
                a["qlen"] = b["qlen"] = qb
                a["tlen"] = b["tlen"] = tl
                a["t_off"] = rb - 1 if tl > 0 else w0
                b["t_off"] = win_off + (tl - 1 if tl > 0 else 0)
                a["flags"], b["flags"] = BMH_F_QREV | BMH_F_TREV | BMH_F_TPAC, BMH_F_QREV | BMH_F_TREV
                a["h0"] = b["h0"] = ql
                a["end_bonus"] = b["end_bonus"] = 5
            else:
                qe = qb + ql
                if qe >= L:
                    continue
                a["q_off"] = b["q_off"] = read_off + qe
                a["qlen"] = b["qlen"] = L - qe
                a["tlen"] = b["tlen"] = w1 - (rb + ql)
                a["t_off"], b["t_off"] = rb + ql, win_off + (rb + ql - w0)
                a["flags"], b["flags"] = BMH_F_TPAC, 0
                a["h0"] = b["h0"] = ql + int(rng.integers(0, 40))
                a["end_bonus"] = b["end_bonus"] = 5
            a["w"] = b["w"] = int(rng.choice([100, 100, 30, 7]))
            tp.append(a), tb.append(b)
    return pac, np.concatenate(pool + [np.zeros(16, np.uint8)]), np.array(tp), np.array(tb)


@pytest.mark.parametrize("mode,read_len", [("lane", (60, 250)), ("lanex4", (250, 560)), ("reg", (60, 400)),
                                           ("grp", (60, 300)), ("lds", (60, 400))])
def test_tpac_tasks_equal_pool_tasks_and_oracle(mode, read_len):
    rng = np.random.default_rng(91)
    l_pac = 100003  # not a multiple of 4: exercises the last partial pac byte
    pac, pool, tp, tb = _pac_tasks(rng, 1500, l_pac, read_len)
    p = kswlib.make_params()
    ctx = _ctx_with({"BMH_EXT_MODE": mode})
    with pytest.raises(Exception):  # TPAC without a reference is an argument error, not a fault
        ctx.extend_batch(pool, tp)
    ctx.set_pac(pac, l_pac)
    want_b, _ = kswlib.orc_extend_batch(p, pool, tb, nthreads=4)
    want_p, _ = kswlib.orc_extend_batch(p, pool, tp, nthreads=4, pac=pac, l_pac=l_pac)
    assert (want_b == want_p).all()  # the checker agrees with itself on the two addressings
    got_p = ctx.extend_batch(pool, tp)
    got_b = ctx.extend_batch(pool, tb)
    assert (got_b == want_b).all()
    bad = np.nonzero(got_p != want_p)[0]
    assert len(bad) == 0, f"{mode}: task {tp[bad[0]]}: gpu={got_p[bad[0]]} oracle={want_p[bad[0]]}"
    # a task that runs off either end of the doubled coordinate is refused
    t = tp[:1].copy()
    t["flags"], t["t_off"], t["tlen"] = BMH_F_TPAC, 2 * l_pac - 3, 10
    with pytest.raises(Exception):
        ctx.extend_batch(pool, t)
    ctx.close()


def test_chain2aln_driver_with_resident_reference_matches_reference_fixture():
    """Same fixture as test_golden_gpu, but the driver emits BMH_F_TPAC tasks and ships no windows."""
    ctx = _ctx_with({})
    nreg = 0
    for p, l_pac, pac, reads, chains, exp in kswlib.golden_chain2aln_groups():
        ctx.set_params(p)
        pac = ctx.set_pac(pac, l_pac)
        got = ctx.chain2aln_batch(l_pac, pac, reads, chains)
        st = ctx.driver_stats()
        for r, (a, b) in enumerate(zip(got, exp)):
            assert len(a) == len(b) and (a == b).all(), f"read {r}: gpu={a} ref={b}"
            nreg += len(b)
        assert st["pool_bytes"] == sum(len(x) for x in reads) + 16, "windows must not be shipped"
    assert nreg >= 2500
    ctx.close()


# ---- the region record: bwa_gen_cigar2's byte work on the device (bmh_region_cigar_batch) ---------------------------------------

def _md_of(words, q, t, rev):
    """bwa.c:134-164 on oriented copies"""
    b2c = b"TGCAN" if rev else b"ACGTN"
    x = y = u = mm = gap = 0
    out = bytearray()
    for k, wd in enumerate(words):
        op, ln = int(wd) & 0xf, int(wd) >> 4
        if op == 0:
            for i in range(ln):
                if q[x + i] != t[y + i]:
                    out += str(u).encode() + b2c[t[y + i]:t[y + i] + 1]
                    mm, u = mm + 1, 0
                else:
                    u += 1
            x, y = x + ln, y + ln
        elif op == 2:
            if 0 < k < len(words) - 1:
                out += str(u).encode() + b"^" + bytes(b2c[c] for c in t[y:y + ln])
                u, gap = 0, gap + ln
            y += ln
        elif op == 1:
            x, gap = x + ln, gap + ln
    out += str(u).encode()
    return mm + gap, bytes(out)


@pytest.mark.parametrize("l_pac", [60000, 2_300_000_011], ids=["small", "coordinates past 2^32"])
def test_region_records_against_oracle_global_and_md(l_pac):
    """bmh_region_cigar_batch record by record: windows of both strands fetched and oriented on the device, three tries per region with
    bands given here, one-try regions, no-gap regions; against the oracle's ksw_global2 on oriented copies made in numpy, the band loop
    replayed in Python and a Python restatement of bwa.c:134-164.  Then the capacity flags, and the loud error without a reference.
    Second case: a 2.3 Gbp reference (575 MB of 2-bit codes), so that forward-strand positions pass 2^31 and reverse-strand ones 2^32."""
    from __graft_entry__ import load_package
    pkg = load_package()
    rng = np.random.default_rng(611)
    pac = rng.integers(0, 256, l_pac // 4 + 1, dtype=np.uint8)

    def window(pos, n):
        """n bases of the doubled coordinate from pos on (bntseq.c:355-376), decoded from the 2-bit array"""
        if pos >= l_pac:  # reverse strand: complement of the forward strand read backwards
            f0 = 2 * l_pac - pos - n
            return (3 - _fwd(f0, n))[::-1]
        return _fwd(pos, n)

    def _fwd(f0, n):
        idx = np.arange(f0, f0 + n, dtype=np.int64)
        return ((pac[idx >> 2] >> ((~idx & 3) << 1).astype(np.uint8)) & 3).astype(np.uint8)

    p = kswlib.make_params()
    ctx = _ctx_with({})
    try:
        ctx.set_params(p)
        reqs, tasks, rpool, opool = [], [], [], []
        rb_, ob_, slot = 0, 0, 0
        INT_MIN = -2 ** 31
        for k in range(400):
            L = int(rng.integers(30, 260))
            rev = k & 1
            lo, hi = (l_pac, 2 * l_pac) if rev else (0, l_pac)
            pos = int(rng.integers(max(lo + 10, hi - 5_000_000), hi - L - 40))  # (near the top of the strand: the largest coordinates)
            tl = L
            read = window(pos, L).copy()
            kind = k % 4
            if kind != 3:  # mismatches, and for kinds 1-2 an indel
                mut = rng.random(L) < 0.05
                read[mut] = (read[mut] + rng.integers(1, 4, mut.sum())) & 3
            if kind in (1, 2) and L > 40:
                c = int(rng.integers(10, L - 10))
                d = int(rng.integers(1, 6))
                if kind == 1:
                    read = np.concatenate([read[:c], read[c + d:]])          # deletion from the read
                else:
                    read = np.concatenate([read[:c], rng.integers(0, 4, d).astype(np.uint8), read[c:]])  # insertion
            if k % 17 == 0:
                read[int(rng.integers(0, len(read)))] = 4  # an N
            ql = len(read)
            q = np.zeros((), pkg.REGION_REQ)
            q["q_src"], q["rb"], q["o_off"], q["ql"], q["tl"] = rb_, pos, ob_, ql, tl
            single = k % 5 == 0
            q["truesc"] = INT_MIN if single else ql - int(rng.integers(0, 30))
            oq = read[::-1] if rev else read
            ot = window(pos, tl)[::-1] if rev else window(pos, tl)
            task = [-1, -1, -1]
            if not (kind in (0, 3) and k % 8 < 4):  # (mismatch-only regions: half of them as the no-gap case)
                prev = -1
                for t_ in range(1 if single else 3):
                    w = max(abs(tl - ql) + 3, (2 + k % 7) << t_)
                    w = min(w, 40)
                    if w == prev:
                        task[t_] = task[t_ - 1]
                        continue
                    prev = w
                    g = np.zeros((), pkg.GLB_TASK)
                    g["q_off"], g["t_off"], g["qlen"], g["tlen"], g["w"], g["cigar_off"], g["cigar_cap"] = ob_, ob_ + ql, ql, tl, w, slot, 24
                    slot += 24
                    task[t_] = len(tasks)
                    tasks.append(g)
            q["task"] = task
            reqs.append(q), rpool.append(read)
            opool += [oq, ot]
            rb_ += ql
            ob_ += ql + tl
        reqs, tasks = np.array(reqs), np.array(tasks)
        rpool, opool = np.concatenate(rpool), np.concatenate(opool)
        with pytest.raises(pkg.BmhError):  # no resident reference yet
            ctx.region_cigar_batch(rpool, len(opool), reqs, tasks, slot + 4)
        ctx.set_pac(pac, l_pac)
        res, cig, md = ctx.region_cigar_batch(rpool, len(opool), reqs, tasks, slot + 4)
        ores, ocig = kswlib.orc_global_batch(p, opool, tasks) if len(tasks) else (None, None)
        a = int(p["a"])
        mat = np.array(p["mat"], dtype=np.int64).reshape(5, 5)
        seen = {"nodp": 0, "tries2": 0, "rev_indel": 0}
        for k, (q, r) in enumerate(zip(reqs, res)):
            ql, tl, rev = int(q["ql"]), int(q["tl"]), int(q["rb"]) >= l_pac
            oq = opool[int(q["o_off"]):int(q["o_off"]) + ql]
            ot = opool[int(q["o_off"]) + ql:int(q["o_off"]) + ql + tl]
            single = int(q["truesc"]) == INT_MIN
            if q["task"][0] < 0:
                sc = int(mat[ot, oq].sum())
                words, tries = np.array([ql << 4], dtype=np.uint32), (2 if (not single and sc < int(q["truesc"]) - a) else 1)
                seen["nodp"] += 1
            else:
                last, tries = -(1 << 30), 0
                for t_ in range(3):
                    fin = int(q["task"][t_])
                    tries += 1
                    sc = int(ores[fin]["score"])
                    if sc == last:
                        break
                    last = sc
                    if single or not (tries < 3 and sc < int(q["truesc"]) - a):
                        break
                n = int(ores[fin]["n_cigar"])
                assert n <= 24
                words = ocig[fin]
                seen["tries2"] += tries >= 2
                seen["rev_indel"] += rev and n > 1
            nm, mds = _md_of(words, oq, ot, rev)
            assert (int(r["score"]), int(r["n_cigar"]), int(r["tries"])) == (sc, len(words), tries), (k, q, r)
            assert int(r["flags"]) == 0 and int(r["NM"]) == nm and int(r["md_len"]) == len(mds), (k, q, r, mds)
            assert np.array_equal(cig[k, :len(words)], words) and bytes(md[k, :len(mds)]) == mds
        assert min(seen.values()) >= 10, seen
        # a batch of no-gap regions only (no ksw_global2 task at all), and an empty batch
        nd = np.nonzero(reqs["task"][:, 0] < 0)[0]
        r3, c3, m3 = ctx.region_cigar_batch(rpool, len(opool), reqs[nd], np.zeros(0, pkg.GLB_TASK), 4)
        assert np.array_equal(r3, res[nd]) and np.array_equal(c3[:, 0], cig[nd, 0])
        assert all(bytes(m3[j, :int(r3[j]['md_len'])]) == bytes(md[k, :int(res[k]['md_len'])]) for j, k in enumerate(nd))  # (bytes past md_len: unspecified)
        r4, _, _ = ctx.region_cigar_batch(rpool, len(opool), reqs[:0], tasks, slot + 4)
        assert len(r4) == 0
        # arguments that would make a kernel read outside its buffers are refused on the host, loudly
        for field, value in (("q_src", len(rpool)), ("o_off", len(opool)), ("rb", 2 * l_pac - 5), ("rb", l_pac - 7), ("task", [len(tasks), -1, -1])):
            bad = reqs.copy()
            k = 1 if field != "task" else int(np.nonzero(reqs["task"][:, 0] >= 0)[0][0])
            bad[field][k] = value
            with pytest.raises(pkg.BmhError):
                ctx.region_cigar_batch(rpool, len(opool), bad, tasks, slot + 4)
        bad_t = tasks.copy()
        bad_t["t_off"][0] = len(opool)
        with pytest.raises(pkg.BmhError):
            ctx.region_cigar_batch(rpool, len(opool), reqs, bad_t, slot + 4)
        # capacities: 2 CIGAR words / 6 MD bytes per region -- what does not fit is flagged, the rest is as before
        res2, cig2, md2 = ctx.region_cigar_batch(rpool, len(opool), reqs, tasks, slot + 4, cig_cap=2, md_cap=6)
        cut = {1: 0, 2: 0, 0: 0}
        for k, (r, r2) in enumerate(zip(res, res2)):
            assert (int(r2["score"]), int(r2["n_cigar"]), int(r2["tries"])) == (int(r["score"]), int(r["n_cigar"]), int(r["tries"]))
            if int(r["n_cigar"]) > 2:
                assert int(r2["flags"]) == pkg.BMH_REGION_CIGAR_CUT
            else:
                assert int(r2["NM"]) == int(r["NM"]) and int(r2["md_len"]) == int(r["md_len"])
                assert int(r2["flags"]) == (pkg.BMH_REGION_MD_CUT if int(r["md_len"]) > 6 else 0)
                if not r2["flags"]:
                    assert bytes(md2[k, :int(r2["md_len"])]) == bytes(md[k, :int(r["md_len"])])
            cut[int(r2["flags"])] += 1
        assert min(cut.values()) >= 10, cut
    finally:
        ctx.close()
