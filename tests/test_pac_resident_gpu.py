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
