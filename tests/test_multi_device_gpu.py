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
