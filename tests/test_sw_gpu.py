"""SURVEY.md §8(f) row 2 -- local Smith-Waterman for mate rescue / short chains on the GPU: bmh_sw_batch must return
kswr_t field for field what the reference's ksw_align2 (ksw.c:341-364) returns, byte-mode / word-mode layout
effects included."""
import numpy as np
import pytest

import kswgen
import kswlib
from __graft_entry__ import load_package
from test_kernel_families_gpu import _ctx_with

pytestmark = pytest.mark.gpu


# the three kernel families behind bmh_sw_batch: one wave per task (what batches of up to 32 k tasks get), one lane per task
# (the register kernels; forced here by turning the wave kernel off), the slab kernel for everything
MODES = ["wave", "lane", "generic"]


def _sw_ctx(mode):
    return _ctx_with({"wave": {"BMH_SW_MODE": "default"}, "lane": {"BMH_SW_MODE": "default", "BMH_SW_WAVE": "0"},
                      "generic": {"BMH_SW_MODE": "generic"}}[mode])


def _cmp(got, want, tasks, what):
    for f in kswlib.SW_FIELDS:
        bad = np.nonzero(got[f] != want[f])[0]
        assert len(bad) == 0, f"{what}, {f}: task {tasks[bad[0]]} gpu={got[bad[0]]} want={want[bad[0]]}"


@pytest.mark.parametrize("mode", MODES)
def test_sw_matches_reference_fixture(mode):
    ctx = _sw_ctx(mode)
    g = kswlib.load_golden("sw_golden.npz")
    pool, tasks, exp, grp, params = g["pool"], g["tasks"], g["expect"], g["group"], g["params"]
    for k in range(len(params)):
        sel = np.nonzero(grp == k)[0]
        ctx.set_params(params[k])
        _cmp(ctx.sw_batch(pool, tasks[sel]), exp[sel], tasks[sel], f"{mode} set {k}")
    ctx.close()


@pytest.mark.parametrize("mode", MODES)
def test_sw_matches_oracle_materescue_and_fuzz(mode):
    ctx = _sw_ctx(mode)
    rng = np.random.default_rng(121)
    p = kswlib.make_params()
    ctx.set_params(p)
    for kw in (dict(n=3000), dict(n=1500, read_len=(30, 160), win=(20, 500), hard=True),
               dict(n=600, read_len=(100, 400), win=(100, 900), hard=True)):
        n = kw.pop("n")
        pool, tasks = kswgen.gen_sw_materescue(rng, n, p, **kw)
        want, _ = kswlib.orc_sw_batch(p, pool, tasks, nthreads=8)
        _cmp(ctx.sw_batch(pool, tasks), want, tasks, f"{mode} materescue {kw}")
    for p in kswgen.sw_param_sets(rng, 8):
        ctx.set_params(p)
        pool, tasks = kswgen.gen_sw_fuzz(rng, 500, p)
        want, _ = kswlib.orc_sw_batch(p, pool, tasks, nthreads=8)
        assert (want["rsv"] == 0).all()
        _cmp(ctx.sw_batch(pool, tasks), want, tasks, f"{mode} fuzz")
    ctx.close()


@pytest.mark.parametrize("mode", MODES)
def test_sw_250bp_class(mode):
    """Queries of 161-256 columns: byte mode below 250 columns, WORD mode (ksw_i16, 8 segments) from there on -- what
    mem_matesw sends for 2 x 250 bp reads (reference bwamem_pair.c:147).  Both go to the 256-column register kernels."""
    ctx = _sw_ctx(mode)
    rng = np.random.default_rng(1250)
    for p, lens in ((kswlib.make_params(), (161, 256)), (kswlib.make_params(), (250, 250)), (kswlib.make_params(), (249, 251)),
                    (kswlib.make_params(a=2, b=5), (100, 256)), (kswlib.make_params(o_del=4, e_del=2, o_ins=6, e_ins=1), (200, 256))):
        ctx.set_params(p)
        for hard in (False, True):
            pool, tasks = kswgen.gen_sw_materescue(rng, 700, p, read_len=lens, win=(50, 600), hard=hard)
            want, _ = kswlib.orc_sw_batch(p, pool, tasks, nthreads=8)
            _cmp(ctx.sw_batch(pool, tasks), want, tasks, f"{mode} {lens} hard={hard}")
    # word mode forced on short queries (the function-level contract: any size with or without KSW_XBYTE)
    p = kswlib.make_params()
    ctx.set_params(p)
    pool, tasks = kswgen.gen_sw_materescue(rng, 1500, p, read_len=(3, 256), win=(0, 300), hard=True)
    tasks["xtra"] &= ~np.uint32(kswlib.KSW_XBYTE)
    want, _ = kswlib.orc_sw_batch(p, pool, tasks, nthreads=8)
    _cmp(ctx.sw_batch(pool, tasks), want, tasks, f"{mode} forced word mode")
    ctx.close()


def test_sw_target_from_resident_reference():
    """BMH_F_TPAC: the rescue window is read straight from the 2-bit reference (what mem_matesw gets from
    bns_get_seq, reference bwamem_pair.c:143), the mate reverse-complemented by flags (bwamem_pair.c:130-133)."""
    from test_pac_resident_gpu import _pac_tasks  # reuses its synthetic genome builder
    rng = np.random.default_rng(131)
    l_pac = 60001
    pac, _, _, _ = _pac_tasks(rng, 1, l_pac, (100, 100))
    bases = ((pac[np.arange(l_pac) >> 2] >> ((~np.arange(l_pac) & 3) << 1)) & 3).astype(np.uint8)
    dbl = np.concatenate([bases, (3 - bases)[::-1]])
    p = kswlib.make_params()
    pool_parts, tasks, off = [], [], 0
    for _ in range(1200):
        L = int(rng.integers(70, 151))
        W = int(rng.integers(200, 600)) + L
        rb = int(rng.integers(0, 2 * l_pac - W))
        if rb < l_pac < rb + W:
            rb = l_pac  # a window never straddles the strand boundary (bns_get_seq returns len 0 then)
        win = dbl[rb:rb + W]
        st = int(rng.integers(0, W - L))
        mate = kswgen.mutate(rng, win[st:st + L], sub=0.03, ins=0.003, dele=0.003, max_indel=3)
        if len(mate) < 20:
            continue
        # store the mate as the reverse complement of what is aligned; QREV|QCOMP undo it
        stored = (3 - mate)[::-1].astype(np.uint8)
        pool_parts.append(stored)
        t = np.zeros((), kswlib.SW_TASK)
        t["q_off"], t["qlen"] = off + len(stored) - 1, len(mate)
        t["t_off"], t["tlen"] = rb, W
        t["flags"] = kswlib.BMH_F_QREV | kswlib.BMH_F_QCOMP | kswlib.BMH_F_TPAC
        t["xtra"] = kswgen.sw_xtra_bwa(p, len(mate))
        tasks.append(t)
        off += len(stored)
    pool = np.concatenate(pool_parts + [np.zeros(16, np.uint8)])
    tasks = np.array(tasks)
    want, _ = kswlib.orc_sw_batch(p, pool, tasks, nthreads=8, pac=pac, l_pac=l_pac)
    assert (want["score"] > 40).mean() > 0.8
    ctx = _ctx_with({})
    with pytest.raises(Exception):
        ctx.sw_batch(pool, tasks)
    ctx.set_pac(pac, l_pac)
    _cmp(ctx.sw_batch(pool, tasks), want, tasks, "tpac")
    ctx.close()


def test_sw_refuses_what_it_cannot_do_exactly():
    pkg = load_package()
    ctx = _ctx_with({})
    rng = np.random.default_rng(141)
    p = kswlib.make_params(o_ins=0)
    ctx.set_params(p)
    pool, tasks = kswgen.gen_sw_materescue(rng, 10, p)
    with pytest.raises(pkg.BmhError):  # o_ins == 0: the reference's lazy-F loop is not a closed recurrence
        ctx.sw_batch(pool, tasks)
    p = kswlib.make_params(a=3, b=6)
    ctx.set_params(p)
    q = kswgen.rand_seq(rng, 100)
    pool = np.concatenate([q, q, np.zeros(8, np.uint8)])
    t = np.zeros(1, kswlib.SW_TASK)
    t["q_off"], t["t_off"], t["qlen"], t["tlen"] = 0, 100, 100, 100
    t["xtra"] = kswlib.KSW_XBYTE | kswlib.KSW_XSUBO | 19  # byte mode overflows: score 255, nothing else (ksw.c:198-200)
    want, _ = kswlib.orc_sw_batch(p, pool, t)
    assert want["score"][0] == 255 and want["qe"][0] == -1
    _cmp(ctx.sw_batch(pool, t), want, t, "byte overflow")
    t["xtra"] |= kswlib.KSW_XSTART  # ... and asking for start positions on top of it is undefined in the reference
    assert kswlib.orc_sw_batch(p, pool, t)[0]["rsv"][0] == 1
    with pytest.raises(pkg.BmhError):
        ctx.sw_batch(pool, t)
    ctx.close()


def test_sw_edge_sizes():
    """Empty targets, single-base queries, and tasks far beyond the register kernels (long queries in word mode,
    long targets) -- the catch-all kernel must agree with the oracle there too."""
    rng = np.random.default_rng(151)
    p = kswlib.make_params()
    pb = kswgen.PoolBuilder(kswlib.SW_TASK)
    X = kswlib.KSW_XSUBO | kswlib.KSW_XSTART | 19
    for qlen, tlen, xtra in [(1, 0, X), (1, 1, X | kswlib.KSW_XBYTE), (150, 0, X | kswlib.KSW_XBYTE), (1, 500, 0),
                             (16, 16, X | kswlib.KSW_XBYTE), (17, 33, X), (160, 700, X | kswlib.KSW_XBYTE),
                             (161, 700, X | kswlib.KSW_XBYTE), (249, 900, X | kswlib.KSW_XBYTE), (250, 900, X),
                             (1200, 5000, X), (3000, 12000, kswlib.KSW_XSTART), (40, 30000, X | kswlib.KSW_XBYTE)]:
        t = kswgen.rand_seq(rng, tlen)
        q = kswgen.rand_seq(rng, qlen)
        if tlen > qlen + 10:
            st = int(rng.integers(0, tlen - qlen))
            q = kswgen.mutate(rng, t[st:st + qlen + 8], sub=0.04, ins=0.004, dele=0.004, max_indel=4)[:qlen]
            if len(q) < qlen:
                q = np.concatenate([q, kswgen.rand_seq(rng, qlen - len(q))])
        kswgen._add_sw(pb, rng, q, t, xtra)
    pool, tasks = pb.finish()
    want, _ = kswlib.orc_sw_batch(p, pool, tasks, nthreads=8)
    assert (want["rsv"] == 0).all()
    ctx = _ctx_with({})
    _cmp(ctx.sw_batch(pool, tasks), want, tasks, "edge sizes")
    ctx.close()


def test_sw_wide_queries_in_a_large_batch():
    """More than 32 768 tasks (so the lane-per-task kernels serve the batch) with queries of up to 300 columns: what exceeds the
    register kernels' 256 columns goes to the one-wave-per-task kernel over the dispatcher's list, not to the slab kernel."""
    ctx = _sw_ctx("wave")
    rng = np.random.default_rng(1301)
    p = kswlib.make_params()
    ctx.set_params(p)
    pool, tasks = kswgen.gen_sw_materescue(rng, 36000, p, read_len=(120, 300), win=(150, 700), hard=True)
    assert int((tasks["qlen"] > 256).sum()) > 2000 and int((tasks["qlen"] <= 160).sum()) > 2000
    want, _ = kswlib.orc_sw_batch(p, pool, tasks, nthreads=8)
    _cmp(ctx.sw_batch(pool, tasks), want, tasks, "large batch with wide queries")
    ctx.close()
