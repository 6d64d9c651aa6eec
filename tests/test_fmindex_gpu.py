"""FM-index queries on the GPU (bmh_smem_batch, bmh_sa_batch) against the reference's outputs (fixture) and the oracle."""
import numpy as np
import pytest

import kswgen
import kswlib
from test_fmindex_cpu import _same_calls
from test_kernel_families_gpu import _ctx_with

pytestmark = pytest.mark.gpu


# both kernels behind bmh_smem_batch (the library picks by batch size): one extension site for all lanes / a loop nest per lane
KERNELS = ["conv", "loops"]


@pytest.mark.parametrize("kernel", KERNELS)
def test_smem_and_sa_match_reference_fixture(kernel, monkeypatch):
    monkeypatch.setenv("BMH_SMEM_KERNEL", kernel)
    cb, keep, raw, opt, reads, per, sa_k, sa_pos = kswlib.golden_fmindex()
    ctx = _ctx_with({})
    with pytest.raises(Exception):  # no index on the device yet
        ctx.sa_batch(sa_k[:4])
    ctx.set_bwt(*raw)
    assert (ctx.sa_batch(sa_k) == sa_pos).all()
    got = ctx.smem_batch(opt, reads)
    for r, (g, w) in enumerate(zip(got, per)):
        assert _same_calls(g, w), f"read {r}: gpu calls {g[0]} want {w[0]}"
    ctx.close()


@pytest.mark.parametrize("kernel", KERNELS)
def test_smem_matches_oracle_on_many_reads(kernel, monkeypatch):
    """More and longer reads than the fixture holds (ragged lengths, N runs, empty reads), against the oracle."""
    monkeypatch.setenv("BMH_SMEM_KERNEL", kernel)
    cb, keep, raw, opt, reads, per, sa_k, sa_pos = kswlib.golden_fmindex()
    rng = np.random.default_rng(191)
    # reads cut from the fixture's own reads (they come from the indexed genome), re-mutated and recombined
    src = np.concatenate(reads)
    more = []
    for _ in range(3000):
        L = int(rng.choice([0, 1, 18, 19, 20, 75, 150, 151, 300, 600]))
        if L == 0:
            more.append(np.zeros(0, np.uint8))
            continue
        p = int(rng.integers(0, len(src) - L))
        rd = src[p:p + L].copy()
        m = rng.random(L) < 0.02
        rd[m] = (rd[m] + rng.integers(1, 4, m.sum())) % 5
        more.append(rd)
    ctx = _ctx_with({})
    ctx.set_bwt(*raw)
    got = ctx.smem_batch(opt, more)
    for r, (g, rd) in enumerate(zip(got, more)):
        w = kswlib.orc_smem_calls(cb, opt, rd) if len(rd) else (np.zeros(0, kswlib.SMEM_CALL), np.zeros(0, kswlib.SMEM_INTV))
        assert _same_calls(g, w), f"read {r} (len {len(rd)})"
    ks = rng.integers(0, raw[2] + 1, 20000).astype(np.uint64)
    assert (ctx.sa_batch(ks) == kswlib.orc_sa(cb, ks)).all()
    ctx.close()


@pytest.mark.parametrize("kernel", KERNELS)
def test_smem_min_emit_len_returns_the_long_intervals_only(kernel, monkeypatch):
    """What the preload shim asks for: the same calls, only the intervals chaining can use (length >= min_seed_len)."""
    monkeypatch.setenv("BMH_SMEM_KERNEL", kernel)
    cb, keep, raw, opt, reads, per, sa_k, sa_pos = kswlib.golden_fmindex()
    o2 = np.array(opt, dtype=kswlib.SMEM_OPT).copy()
    o2["min_emit_len"] = int(o2["min_seed_len"])
    rng = np.random.default_rng(193)
    src = np.concatenate(reads)
    more = list(reads)
    for _ in range(2000):
        L = int(rng.choice([1, 19, 20, 75, 150, 151, 300]))
        p = int(rng.integers(0, len(src) - L))
        rd = src[p:p + L].copy()
        m = rng.random(L) < rng.choice([0.02, 0.12])
        rd[m] = (rd[m] + rng.integers(1, 4, m.sum())) % 5
        more.append(rd)
    ctx = _ctx_with({})
    ctx.set_bwt(*raw)
    got = ctx.smem_batch(o2, more)
    full = ctx.smem_batch(opt, more)
    n_kept = n_all = 0
    for r, (g, rd) in enumerate(zip(got, more)):
        assert _same_calls(g, kswlib.orc_smem_calls(cb, o2, rd)), f"read {r} (len {len(rd)})"
        n_kept += len(g[1])
        n_all += len(full[r][1])
    assert 0 < n_kept < n_all / 4
    ctx.close()


@pytest.mark.parametrize("kernel", KERNELS)
def test_seed_batch_is_smem_batch_plus_the_suffix_array_lookups(kernel, monkeypatch):
    """bmh_seed_batch: the intervals of bmh_smem_batch and, for each one chaining would look up (length >= min_seed_len,
    x[2] <= max_occ), its suffix-array positions -- the same numbers as bmh_sa_batch / the oracle's bwt_sa, in one call."""
    monkeypatch.setenv("BMH_SMEM_KERNEL", kernel)
    cb, keep, raw, opt, reads, per, sa_k, sa_pos = kswlib.golden_fmindex()
    rng = np.random.default_rng(197)
    src = np.concatenate(reads)
    more = list(reads)
    for _ in range(1500):
        L = int(rng.choice([19, 20, 75, 150, 151, 300]))
        p = int(rng.integers(0, len(src) - L))
        rd = src[p:p + L].copy()
        m = rng.random(L) < rng.choice([0.0, 0.02, 0.12])
        rd[m] = (rd[m] + rng.integers(1, 4, m.sum())) % 5
        more.append(rd)
    ctx = _ctx_with({})
    ctx.set_bwt(*raw)
    for emit, max_occ in ((0, 10000), (int(opt["min_seed_len"]), 10000), (int(opt["min_seed_len"]), 3)):
        o2 = np.array(opt, dtype=kswlib.SMEM_OPT).copy()
        o2["min_emit_len"] = emit
        got, offs, pos = ctx.seed_batch(o2, max_occ, more)
        plain = ctx.smem_batch(o2, more)
        n_looked = n_skipped = 0
        want_keys, got_pos = [], []
        for r, ((gc, gi), (pc, pi), so) in enumerate(zip(got, plain, offs)):
            assert _same_calls((gc, gi), (pc, pi)), f"read {r}"
            ln = (gi["info"] & np.uint64(0xffffffff)).astype(np.int64) - (gi["info"] >> np.uint64(32)).astype(np.int64)
            look = (ln >= int(opt["min_seed_len"])) & (gi["x2"] <= max_occ)
            assert ((so != np.uint64(0xffffffffffffffff)) == look).all(), f"read {r}"
            n_looked += int(look.sum())
            n_skipped += int((~look).sum())
            for k in np.nonzero(look)[0]:
                x0, x2, b = int(gi["x0"][k]), int(gi["x2"][k]), int(so[k])
                want_keys.append(np.arange(x0, x0 + x2, dtype=np.uint64))
                got_pos.append(pos[b:b + x2])
        assert n_looked > 1000 and (emit or n_skipped > 1000)
        keys = np.concatenate(want_keys)
        assert len(keys) == len(pos)  # every position slot belongs to exactly one interval
        assert (np.concatenate(got_pos) == ctx.sa_batch(keys)).all()
        sample = rng.choice(len(keys), 300, replace=False)
        assert (np.concatenate(got_pos)[sample] == kswlib.orc_sa(cb, keys[sample])).all()
    ctx.close()


@pytest.mark.parametrize("kernel", KERNELS)
def test_seed_batch_survives_an_overflowed_first_attempt(kernel, monkeypatch):
    """Output arrays far too small for the first attempt (BMH_SMEM_INIT_CAP): the SMEM kernel overflows, the look-up kernel behind it
    must not walk the unwritten slots (stale bytes of a buffer other stages share -- run those stages first so that it IS dirty), the
    host retries with arrays that fit, and the results are those of a comfortable first attempt."""
    monkeypatch.setenv("BMH_SMEM_KERNEL", kernel)
    cb, keep, raw, opt, reads, per, sa_k, sa_pos = kswlib.golden_fmindex()
    rng = np.random.default_rng(5)
    src = np.concatenate(reads)
    more = list(reads) + [src[p:p + 150].copy() for p in rng.integers(0, len(src) - 150, 800)]
    o2 = np.array(opt, dtype=kswlib.SMEM_OPT).copy()
    ctx = _ctx_with({})
    ctx.set_bwt(*raw)
    want, woffs, wpos = ctx.seed_batch(o2, 10000, more)
    ctx.close()
    monkeypatch.setenv("BMH_SMEM_INIT_CAP", "64")
    ctx = _ctx_with({})
    ctx.set_bwt(*raw)
    # dirty the shared scratch with all-ones intervals: x2 <= max_occ is false for them only if the guard looks at them at all;
    # what matters is that the call returns, with the same answers
    p = kswlib.make_params()
    pool, st = kswgen.gen_sw_materescue(rng, 4000, p)
    ctx.sw_batch(pool, st)
    got, offs, pos = ctx.seed_batch(o2, 10000, more)
    assert len(pos) == len(wpos)
    for r, ((gc, gi), (wc, wi), so, wo) in enumerate(zip(got, want, offs, woffs)):
        assert _same_calls((gc, gi), (wc, wi)), f"read {r}"
        look = so != np.uint64(0xffffffffffffffff)
        assert (look == (wo != np.uint64(0xffffffffffffffff))).all()
        for k in np.nonzero(look)[0]:
            x2 = int(gi["x2"][k])
            assert (pos[int(so[k]):int(so[k]) + x2] == wpos[int(wo[k]):int(wo[k]) + x2]).all()
    ctx.close()
