"""Multi-GPU readiness without a node: the path shards over independent units with no collective, so what
can be checked on however many GPUs are visible -- one on this pool, eight on a node, the same code either way -- is that
a static contiguous split over ONE CONTEXT PER VISIBLE DEVICE returns exactly what one context returns, at the size of a
bench chunk (1 M reads), for the flat extension batch (bmh_extend_batch_sharded) and for the fused per-seed records."""
import importlib

import numpy as np
import pytest

import kswlib
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu


def test_one_context_per_visible_device_equals_one_context():
    import torch
    pkg = load_package()
    tg = importlib.import_module("bwa_mem_quickassist_amd.taskgen")
    sh = importlib.import_module("bwa_mem_quickassist_amd.shard")
    ndev = torch.cuda.device_count()
    assert ndev >= 1
    p = kswlib.make_params()
    pool, tasks, tread = tg.generate(p, 1_000_000, "150bp", seed=sh.shard_seed(7, 0))
    one = pkg.Context(0, p)
    want = one.extend_batch(pool, tasks)
    ctxs = [pkg.Context(d, p) for d in range(ndev)]
    got = pkg.extend_batch_sharded(ctxs, pool, tasks)
    assert (got == want).all()
    # with more shards than devices too (two contexts per device): the split, not the device count, is what is exercised
    ctxs2 = ctxs + [pkg.Context(d, p) for d in range(ndev)]
    got2 = pkg.extend_batch_sharded(ctxs2, pool, tasks)
    assert (got2 == want).all()
    # a sample against the oracle, so that "equal" is not "equally wrong"
    ow, _ = kswlib.orc_extend_batch(p, pool, tasks[:20000], nthreads=8)
    assert (ow == want[:20000]).all()
    # fused per-seed records: each shard's slice through its own context, concatenated
    spool, seeds = tg.generate_seeds(p, 300_000, "150bp", seed=11)
    whole = one.seedext_batch(spool, seeds)
    parts = []
    for g, (lo, hi) in enumerate(sh.shard_ranges(len(seeds), len(ctxs2))):
        parts.append(ctxs2[g].seedext_batch(spool, seeds[lo:hi]))
    assert (np.concatenate(parts) == whole).all()
    for c in ctxs2 + [one]:
        c.close()


def test_sharded_entry_points_of_the_other_batches():
    """bmh_seedext_batch_sharded / bmh_global_batch_sharded / bmh_sw_batch_sharded: contiguous task ranges over a context list
    (one per visible device, then two per device), everything enqueued before any wait, results at their task index and every
    CIGAR where its task says -- equal to one context, and a sample equal to the oracle."""
    import torch
    pkg = load_package()
    tg = importlib.import_module("bwa_mem_quickassist_amd.taskgen")
    ndev = torch.cuda.device_count()
    p = kswlib.make_params()
    one = pkg.Context(0, p)
    lists = [[pkg.Context(d, p) for d in range(ndev)], [pkg.Context(d % ndev, p) for d in range(2 * ndev + 1)]]
    spool, seeds = tg.generate_seeds(p, 200_000, "150bp", seed=5)
    want_s = one.seedext_batch(spool, seeds)
    gpool, gtasks, gwords = tg.generate_global(120_000, "150bp", seed=6)
    # CIGAR ranges handed out in REVERSE task order: the shards' ranges then interleave in the caller's pool
    order = np.argsort(-gtasks["cigar_off"].astype(np.int64), kind="stable")
    want_g, want_c = one.global_batch(gpool, gtasks, gwords)
    wpool, wtasks = tg.generate_sw(p, 60_000, "150bp", seed=7)
    want_w = one.sw_batch(wpool, wtasks)
    for ctxs in lists:
        assert (pkg.seedext_batch_sharded(ctxs, spool, seeds) == want_s).all()
        got_g, got_c = pkg.global_batch_sharded(ctxs, gpool, gtasks, gwords)
        assert (got_g == want_g).all()
        for k in range(0, len(gtasks), 13):
            o, n = int(gtasks[k]["cigar_off"]), int(want_g[k]["n_cigar"])
            assert (got_c[o:o + n] == want_c[o:o + n]).all()
        shuffled = gtasks[order]
        got_g2, got_c2 = pkg.global_batch_sharded(ctxs, gpool, shuffled, gwords)
        assert (got_g2 == want_g[order]).all()
        for k in range(0, len(shuffled), 17):
            o, n = int(shuffled[k]["cigar_off"]), int(got_g2[k]["n_cigar"])
            assert (got_c2[o:o + n] == want_c[o:o + n]).all()
        got_w = pkg.sw_batch_sharded(ctxs, wpool, wtasks)
        assert all((got_w[f] == want_w[f]).all() for f in kswlib.SW_FIELDS)
    ow, _, _ = kswlib.orc_seedext_batch(p, spool, seeds[:5000], nthreads=8)
    assert all((ow[f] == want_s[:5000][f]).all() for f in ow.dtype.names)
    # an out-of-range task in the LAST shard is an error after the earlier shards were enqueued: they are drained, nothing hangs
    bad = seeds.copy()
    bad["q_off"][-1] = np.uint64(len(spool) + 5)
    with pytest.raises(pkg.BmhError):
        pkg.seedext_batch_sharded(lists[1], spool, bad)
    assert (pkg.seedext_batch_sharded(lists[1], spool, seeds) == want_s).all()
    for c in lists[0] + lists[1] + [one]:
        c.close()
