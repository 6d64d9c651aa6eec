"""Every extension / global kernel family stays bit-exact, not only the default (lane-per-task) path.
BMH_EXT_MODE / BMH_GLB_MODE are read when a context is created."""
import os

import numpy as np
import pytest

import kswgen
import kswlib
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu


def _ctx_with(env):
    pkg = load_package()
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return pkg.Context(0, kswlib.make_params())
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("mode", ["reg", "grp", "lds", "lanex4", "lane"])
def test_extend_family_matches_reference_fixture_and_oracle(mode):
    ctx = _ctx_with({"BMH_EXT_MODE": mode})
    g = kswlib.load_golden("ext_golden.npz")
    pool, tasks, exp, grp, params = g["pool"], g["tasks"], g["expect"], g["group"], g["params"]
    for k in range(len(params)):
        sel = np.nonzero(grp == k)[0]
        ctx.set_params(params[k])
        got = ctx.extend_batch(pool, tasks[sel])
        bad = np.nonzero(got != exp[sel])[0]
        assert len(bad) == 0, f"{mode}: set {k} task {tasks[sel][bad[0]]}: gpu={got[bad[0]]} ref={exp[sel][bad[0]]}"
    # long flanks (130-500 bp) so that the 2- and 4-lanes-per-task / LDS kernels really run
    rng = np.random.default_rng(77)
    p = kswlib.make_params()
    pool, tasks = kswgen.gen_ext_realistic(rng, 600, read_len=(250, 560), hard=True)
    ctx.set_params(p)
    want, _ = kswlib.orc_extend_batch(p, pool, tasks)
    assert (ctx.extend_batch(pool, tasks) == want).all()
    ctx.close()


def test_many_long_flanks_take_the_lanes_per_task_kernel():
    """> 4096 tasks with 129-256 bp queries: the device-side count switches bin 3 to extend_lanex_kernel<2>."""
    ctx = _ctx_with({"BMH_EXT_MODE": "lane"})
    rng = np.random.default_rng(78)
    p = kswlib.make_params()
    pool, tasks = kswgen.gen_ext_realistic(rng, 4000, read_len=(300, 400), hard=False)
    tasks = np.concatenate([tasks] * 6)  # the same pool, six records per sequence pair: enough tasks to flip the switch
    assert ((tasks["qlen"] > 128) & (tasks["qlen"] <= 256)).sum() > 4096
    want, _ = kswlib.orc_extend_batch(p, pool, tasks, nthreads=8)
    assert (ctx.extend_batch(pool, tasks) == want).all()
    ctx.close()


@pytest.mark.parametrize("mode", ["wave", "lane"])
def test_global_family_matches_reference_fixture(mode):
    ctx = _ctx_with({"BMH_GLB_MODE": mode})
    g = kswlib.load_golden("glb_golden.npz")
    pool, tasks, exp, grp, params, cigar = g["pool"], g["tasks"], g["expect"], g["group"], g["params"], g["cigar"]
    for k in range(len(params)):
        sel = np.nonzero(grp == k)[0]
        ctx.set_params(params[k])
        res, cig = ctx.global_batch(pool, tasks[sel], len(cigar))
        assert (res == exp[sel]).all(), f"{mode}: set {k}"
        for t, r in zip(tasks[sel], res):
            o, n = int(t["cigar_off"]), int(r["n_cigar"])
            assert np.array_equal(cig[o:o + n], cigar[o:o + n])
    ctx.close()
