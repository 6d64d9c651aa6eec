"""GPU parity of the fused per-seed extension record (reference bwamem.c:808-866, the record the fork
sketched at :553-577): bmh_seedext_batch through the C-ABI must equal the oracle's per-seed function bit for bit --
including the right extension's h0 (= the left score the DEVICE computed), both band-doubling retries and a->w."""
import importlib

import numpy as np
import pytest

import kswlib
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu

FIELDS = ("qb", "qe", "rb", "re", "score", "truesc", "w", "n_ext")


@pytest.fixture(scope="module")
def pkg():
    return load_package()


@pytest.fixture(scope="module", params=["auto", "lane", "persist"])
def ctx(pkg, request):
    from test_kernel_families_gpu import _ctx_with
    env = {"auto": {}, "lane": {"BMH_EXT_SMALL": "0"}, "persist": {"BMH_EXT_SMALL": "0", "BMH_EXT_PERSIST": "1"}}[request.param]
    c = _ctx_with(env)
    yield c
    c.close()


def _cmp(ctx, p, pool, tasks, pac=None, l_pac=0):
    ctx.set_params(p)
    got = ctx.seedext_batch(pool, tasks)
    want, _, calls = kswlib.orc_seedext_batch(p, pool, tasks, nthreads=8, pac=pac, l_pac=l_pac)
    for f in FIELDS:
        bad = np.nonzero(got[f] != want[f])[0]
        assert len(bad) == 0, f"{len(bad)} seeds differ in {f}; first {bad[0]}: task={tasks[bad[0]]} gpu={got[bad[0]]} oracle={want[bad[0]]}"
    st = ctx.seedext_stats()
    assert st["seeds"] == len(tasks)
    assert st["left_tasks"] + st["left_retries"] + st["right_tasks"] + st["right_retries"] == calls
    return got, st


@pytest.mark.parametrize("w,workload,n", [(100, "150bp", 6000), (100, "mixed100-300", 3000), (100, "250bp", 2000)])
def test_seedext_default_band(pkg, ctx, w, workload, n):
    tg = importlib.import_module("bwa_mem_quickassist_amd.taskgen")
    p = kswlib.make_params(w=w)
    pool, tasks = tg.generate_seeds(p, n, workload, seed=401)
    # twice: the second call runs with the bin-size hints of the first (grid sizes / kernel choice come from them)
    _cmp(ctx, p, pool, tasks)
    _cmp(ctx, p, pool, tasks)


def test_seedext_bins_outgrow_the_hint(pkg, ctx):
    """a small batch, then a batch 20x its size and of another length mix, on the same context: the second runs with grids sized from the
    first one's bin counts, every bin several times larger than its estimate -- the strided pick-up launches, and for the 65-128-column bin
    the hand-over of the 96-column head's remainder to the 128-column launch, do the rest"""
    tg = importlib.import_module("bwa_mem_quickassist_amd.taskgen")
    p = kswlib.make_params(w=100)
    pool, tasks = tg.generate_seeds(p, 1000, "150bp", seed=411)
    _cmp(ctx, p, pool, tasks)
    _cmp(ctx, p, pool, tasks)   # (the hint of a launch arrives with the next one)
    pool, tasks = tg.generate_seeds(p, 20000, "mixed100-300", seed=412)
    _cmp(ctx, p, pool, tasks)
    pool, tasks = tg.generate_seeds(p, 1500, "250bp", seed=413)
    _cmp(ctx, p, pool, tasks)


@pytest.mark.parametrize("w", [8, 14, 25])
def test_seedext_narrow_band_forces_both_retries(pkg, ctx, w):
    tg = importlib.import_module("bwa_mem_quickassist_amd.taskgen")
    p = kswlib.make_params(w=w)
    pool, tasks = tg.generate_seeds(p, 4000, "mixed100-300", seed=402 + w)
    got, st = _cmp(ctx, p, pool, tasks)
    assert st["left_retries"] > 0 and st["right_retries"] > 0
    assert (got["w"] == 2 * w).any() and (got["w"] == w).any()


def test_seedext_option_sets(pkg, ctx):
    tg = importlib.import_module("bwa_mem_quickassist_amd.taskgen")
    for kw in (dict(a=2, b=8, o_del=12, e_del=2, o_ins=12, e_ins=2, w=30, zdrop=200, pen_clip5=10, pen_clip3=10),
               dict(o_del=6, o_ins=4, e_del=1, e_ins=2, w=40), dict(zdrop=20, w=50), dict(zdrop=0, w=60, pen_clip5=0, pen_clip3=9)):
        p = kswlib.make_params(**kw)
        pool, tasks = tg.generate_seeds(p, 2500, "mixed100-300", seed=77)
        _cmp(ctx, p, pool, tasks)


def test_seedext_edges(pkg, ctx):
    """seed at the read start (no left flank), at the read end (no right flank), covering the whole read, empty window
    flanks (rbeg = 0 / no reference right of the seed), one-base flanks."""
    rng = np.random.default_rng(5)
    p = kswlib.make_params()
    L, Wn = 80, 200
    pool = rng.integers(0, 4, 4000, dtype=np.uint8)
    read = pool[1000:1000 + L].copy()
    pool[0:L] = read
    pool[200:200 + Wn] = pool[1000 - 60:1000 - 60 + Wn]  # window: the read's locus with 60 bases of left context
    rows = [(0, 30, 60), (50, 30, 110), (0, 80, 60), (1, 30, 61), (49, 30, 109), (20, 40, 80)]
    tasks = np.zeros(len(rows) + 2, dtype=pkg.SEED_TASK)
    for k, (qb, ln, rb) in enumerate(rows):
        tasks[k] = (0, 200, L, qb, ln, rb, Wn, 0, 0)
    tasks[len(rows)] = (0, 260, L, 25, 30, 0, Wn - 60, 0, 0)             # rbeg = 0: left target empty
    tasks[len(rows) + 1] = (0, 200, L, 20, 30, 80, 110, 0, 0)            # window ends with the seed: right target empty
    _cmp(ctx, p, pool, tasks)


def test_seedext_reference_resident(pkg, ctx):
    """BMH_F_TPAC: windows read from the 2-bit reference in HBM, forward and reverse strand."""
    tg = importlib.import_module("bwa_mem_quickassist_amd.taskgen")
    p = kswlib.make_params(w=30)
    pool, tasks = tg.generate_seeds(p, 2000, "150bp", seed=9)
    l_pac = len(pool)
    q = np.concatenate([pool & 3, np.zeros((-l_pac) % 4 + 4, np.uint8)])
    q = q[: len(q) // 4 * 4].reshape(-1, 4)
    pac = (q[:, 0] << 6 | q[:, 1] << 4 | q[:, 2] << 2 | q[:, 3]).astype(np.uint8)
    tasks = tasks.copy()
    tasks["flags"] |= pkg.BMH_F_TPAC
    # every other seed on the reverse strand: the read is reverse-complemented in a second pool region, the window is the
    # mirror image on the doubled coordinate
    pool2 = np.concatenate([pool, np.zeros(0, np.uint8)])
    extra = []
    base = len(pool2)
    for k in range(0, len(tasks), 2):
        t = tasks[k]
        Lq = int(t["l_query"])
        rd = pool[int(t["q_off"]): int(t["q_off"]) + Lq]
        rc = np.where(rd < 4, 3 - rd[::-1], 4).astype(np.uint8)
        extra.append(rc)
        wl, rm0 = int(t["wlen"]), int(t["t_off"])
        t["q_off"] = base
        base += Lq
        t["t_off"] = 2 * l_pac - (rm0 + wl)                 # window [rm0, rm0+wl) mirrored
        qb, ln, rb = int(t["qbeg"]), int(t["len"]), int(t["rbeg"])
        t["qbeg"] = Lq - qb - ln
        t["rbeg"] = wl - rb - ln
        tasks[k] = t
    pool2 = np.concatenate([pool] + extra + [np.zeros(16, np.uint8)])
    ctx.set_pac(pac, l_pac)
    _cmp(ctx, p, pool2, tasks, pac=pac, l_pac=l_pac)


def test_seedext_refuses_out_of_range(pkg, ctx):
    p = kswlib.make_params()
    ctx.set_params(p)
    pool = np.zeros(1000, dtype=np.uint8)
    t = np.zeros(1, dtype=pkg.SEED_TASK)
    t[0] = (0, 200, 100, 90, 30, 10, 300, 0, 0)  # seed sticks out of the read
    with pytest.raises(pkg.BmhError) as e:
        ctx.seedext_batch(pool, t)
    assert e.value.code == pkg.BMH_E_ARG
