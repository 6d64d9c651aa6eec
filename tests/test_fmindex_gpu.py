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
