"""GPU parity tests proper: the HIP path, called through the C-ABI (ctypes ->
libbwamem_hip.so), must be BIT-EXACT against the CPU oracle on the same seeded
inputs.  Integer work: the tolerance is zero."""
import numpy as np
import pytest

import kswgen
import kswlib
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return load_package()


# Small batches (what these tests send) go to the one-task-per-wave kernels by default; "lane" keeps them on the
# lane-per-task kernels that serve the large batches of the bench, so both production paths see every case below.
# The global lane kernels have one more switch: the unmasked body for blocks inside every lane's band ("masked" turns it off).
@pytest.fixture(scope="module", params=["auto", "lane", "masked"])
def ctx(pkg, request):
    from test_kernel_families_gpu import _ctx_with
    c = _ctx_with({"auto": {}, "lane": {"BMH_EXT_SMALL": "0"}, "masked": {"BMH_EXT_SMALL": "0", "BMH_GL_FAST": "0"}}[request.param])
    yield c
    c.close()


def _cmp_ext(ctx, p, pool, tasks):
    ctx.set_params(p)
    got = ctx.extend_batch(pool, tasks)
    want, _ = kswlib.orc_extend_batch(p, pool, tasks)
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, f"{len(bad)} mismatches; first task {bad[0]}: {tasks[bad[0]]} gpu={got[bad[0]]} oracle={want[bad[0]]}"


def test_extend_realistic_150bp(ctx):
    rng = np.random.default_rng(201)
    pool, tasks = kswgen.gen_ext_realistic(rng, 4000)
    _cmp_ext(ctx, kswlib.make_params(), pool, tasks)


def test_extend_hard_100_300bp(ctx):
    rng = np.random.default_rng(202)
    pool, tasks = kswgen.gen_ext_realistic(rng, 3000, read_len=(100, 300), hard=True)
    _cmp_ext(ctx, kswlib.make_params(), pool, tasks)


def test_extend_long_queries_multichunk(ctx):
    """qlen up to 1000: several 64-column chunks per row, band retry widths, long targets (>256 rows)."""
    rng = np.random.default_rng(203)
    pool, tasks = kswgen.gen_ext_realistic(rng, 300, read_len=(600, 1100), hard=True, w=100)
    _cmp_ext(ctx, kswlib.make_params(), pool, tasks)
    pool, tasks = kswgen.gen_ext_realistic(rng, 200, read_len=(600, 1100), hard=False, w=200)
    _cmp_ext(ctx, kswlib.make_params(w=200, zdrop=0), pool, tasks)


def test_extend_fuzz_param_sets(ctx):
    rng = np.random.default_rng(204)
    for p in kswgen.fuzz_param_sets(rng, 40):
        pool, tasks = kswgen.gen_ext_fuzz(rng, 400, p)
        _cmp_ext(ctx, p, pool, tasks)


def test_extend_empty_and_tiny_batches(ctx):
    p = kswlib.make_params()
    ctx.set_params(p)
    assert len(ctx.extend_batch(np.zeros(8, np.uint8), np.zeros(0, kswlib.EXT_TASK))) == 0
    rng = np.random.default_rng(205)
    pool, tasks = kswgen.gen_ext_realistic(rng, 1)
    _cmp_ext(ctx, p, pool, tasks)


def test_extend_rejects_out_of_range(ctx, pkg):
    p = kswlib.make_params()
    ctx.set_params(p)
    pool = np.zeros(70000, np.uint8)
    t = np.zeros(1, kswlib.EXT_TASK)
    t["qlen"], t["tlen"], t["h0"], t["w"], t["t_off"] = 40000, 10, 100, 100, 40000  # 40000*1+100 > 32000
    with pytest.raises(pkg.BmhError) as e:
        ctx.extend_batch(pool, t)
    assert e.value.code == pkg.BMH_E_RANGE
    t["qlen"], t["q_off"] = 10, 69999  # reads past the pool
    with pytest.raises(pkg.BmhError) as e:
        ctx.extend_batch(pool, t)
    assert e.value.code == pkg.BMH_E_ARG
    with pytest.raises(pkg.BmhError):
        ctx.set_params(kswlib.make_params(e_ins=0))


def test_extend_sharded_two_contexts_one_device(ctx, pkg):
    """The static multi-GPU split (one context per device) exercised with 2 contexts on GPU 0."""
    p = kswlib.make_params()
    c2 = pkg.Context(0, p)
    ctx.set_params(p)
    rng = np.random.default_rng(206)
    pool, tasks = kswgen.gen_ext_realistic(rng, 1001)
    got = pkg.extend_batch_sharded([ctx, c2], pool, tasks)
    want, _ = kswlib.orc_extend_batch(p, pool, tasks)
    assert (got == want).all()
    c2.close()


def _cmp_glb(ctx, p, pool, tasks, words):
    ctx.set_params(p)
    res, cig = ctx.global_batch(pool, tasks, words)
    ores, ocig = kswlib.orc_global_batch(p, pool, tasks)
    bad = np.nonzero(res != ores)[0]
    assert len(bad) == 0, f"first mismatch {bad[0]}: {tasks[bad[0]]} gpu={res[bad[0]]} oracle={ores[bad[0]]}"
    for k, (t, r, oc) in enumerate(zip(tasks, res, ocig)):
        got = cig[int(t["cigar_off"]): int(t["cigar_off"]) + int(r["n_cigar"])]
        assert np.array_equal(got, oc), f"task {k}: cigar gpu={got} oracle={oc}"


def test_global_realistic(ctx):
    rng = np.random.default_rng(207)
    pool, tasks, words = kswgen.gen_glb_realistic(rng, 1500)
    _cmp_glb(ctx, kswlib.make_params(), pool, tasks, words)
    pool, tasks, words = kswgen.gen_glb_realistic(rng, 800, read_len=(100, 300), hard=True)
    _cmp_glb(ctx, kswlib.make_params(), pool, tasks, words)


def test_global_band_wider_than_the_query_alone_in_a_batch(ctx):
    """A task whose band (32..63) is wider than its query is short: the host's sizing looks at min(w, qlen), the device bins on w
    as given -- the 128-slot kernel must be launched for it even when nothing else in the batch lifts the batch's band."""
    rng = np.random.default_rng(2081)
    for w, ql, tl in ((40, 20, 22), (63, 9, 12), (33, 31, 28)):
        q = rng.integers(0, 4, ql, dtype=np.uint8)
        t = np.concatenate([q[: ql // 2], rng.integers(0, 4, tl - ql // 2, dtype=np.uint8)])[:tl]
        pool = np.concatenate([q, t, np.zeros(16, np.uint8)])
        tasks = np.zeros(1, dtype=kswlib.GLB_TASK)
        tasks[0] = (0, ql, ql, tl, w, 0, ql + tl + 2)
        _cmp_glb(ctx, kswlib.make_params(), pool, tasks, ql + tl + 2)


def test_global_long_hbm_scratch(ctx):
    """Direction matrix too large for LDS -> HBM scratch slab variant of the kernel."""
    rng = np.random.default_rng(208)
    pool, tasks, words = kswgen.gen_glb_realistic(rng, 60, read_len=(900, 1500), hard=True)
    _cmp_glb(ctx, kswlib.make_params(), pool, tasks, words)


def test_global_fuzz_param_sets(ctx):
    rng = np.random.default_rng(209)
    for p in kswgen.fuzz_param_sets(rng, 25):
        pool, tasks, words = kswgen.gen_glb_fuzz(rng, 200)
        _cmp_glb(ctx, p, pool, tasks, words)


def test_staged_and_direct_transfers_agree(pkg):
    """Host buffers of up to 64 MB per direction go through the context's pinned staging buffers, larger ones the direct way
    (csrc/api.hip, Stager): the same batch must come back identical either way, and with staging reserved ahead of time."""
    rng = np.random.default_rng(207)
    p = kswlib.make_params()
    pool, tasks = kswgen.gen_ext_realistic(rng, 3000)
    want, _ = kswlib.orc_extend_batch(p, pool, tasks)
    c = pkg.Context(0, p)
    c.reserve_staging(1 << 20, 1 << 20)
    assert (c.extend_batch(pool, tasks) == want).all()                       # staged
    big = np.concatenate([pool, np.zeros(80 << 20, np.uint8)])             # > 64 MB up: not staged
    assert (c.extend_batch(big, tasks) == want).all()
    c.close()


def test_sleeping_waits_give_the_same_results(pkg):
    """bmh_set_wait_mode(1): host threads sleep on a blocking event instead of spinning on the stream -- what the preload
    shim runs with.  Same calls, same results; the mode is process-wide and is put back."""
    from test_kernel_families_gpu import _ctx_with
    rng = np.random.default_rng(207)
    p = kswlib.make_params()
    pool, tasks = kswgen.gen_ext_realistic(rng, 3000)
    want, _ = kswlib.orc_extend_batch(p, pool, tasks)
    gpool, gt, words = kswgen.gen_glb_realistic(rng, 500)
    gwant, _ = kswlib.orc_global_batch(p, gpool, gt)
    lib = pkg.lib()
    lib.bmh_set_wait_mode.restype = None
    lib.bmh_set_wait_mode(1)
    try:
        c = _ctx_with({})  # created in sleeping mode: its device gets hipDeviceScheduleBlockingSync as well
        assert (c.extend_batch(pool, tasks) == want).all()
        res, _ = c.global_batch(gpool, gt, words)
        assert (res == gwant).all()
        c.close()
    finally:
        lib.bmh_set_wait_mode(0)
