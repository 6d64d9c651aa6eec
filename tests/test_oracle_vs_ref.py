"""Pin the CPU restatement (oracle/) against the REFERENCE ITSELF compiled in
this container (oracle/_ref/libksw_ref.so, built from /root/reference by
oracle/Makefile).  Skipped where the compiled reference is absent."""
import numpy as np
import pytest

import kswgen
import kswlib

pytestmark = pytest.mark.skipif(not kswlib.have_ref(), reason="oracle/_ref not built (no /root/reference here)")


def _check_ext(p, pool, tasks):
    ref = kswlib.ref_extend_batch(p, pool, tasks)
    orc, _ = kswlib.orc_extend_batch(p, pool, tasks)
    bad = np.nonzero(ref != orc)[0]
    assert len(bad) == 0, f"first mismatch task {bad[0]}: {tasks[bad[0]]} ref={ref[bad[0]]} orc={orc[bad[0]]}"


def test_extend_realistic_150bp():
    rng = np.random.default_rng(101)
    pool, tasks = kswgen.gen_ext_realistic(rng, 1500)
    _check_ext(kswlib.make_params(), pool, tasks)


def test_extend_hard_100_300bp():
    rng = np.random.default_rng(102)
    pool, tasks = kswgen.gen_ext_realistic(rng, 1500, read_len=(100, 300), hard=True)
    _check_ext(kswlib.make_params(), pool, tasks)


def test_extend_fuzz_param_sets():
    rng = np.random.default_rng(103)
    for p in kswgen.fuzz_param_sets(rng, 40):
        pool, tasks = kswgen.gen_ext_fuzz(rng, 300, p)
        _check_ext(p, pool, tasks)


def _check_glb(p, pool, tasks):
    ref, rc = kswlib.ref_global_batch(p, pool, tasks)
    orc, oc = kswlib.orc_global_batch(p, pool, tasks)
    assert (ref == orc).all()
    for a, b in zip(rc, oc):
        assert np.array_equal(a, b)


def test_global_realistic():
    rng = np.random.default_rng(104)
    pool, tasks, _ = kswgen.gen_glb_realistic(rng, 800)
    _check_glb(kswlib.make_params(), pool, tasks)
    pool, tasks, _ = kswgen.gen_glb_realistic(rng, 500, read_len=(100, 300), hard=True)
    _check_glb(kswlib.make_params(), pool, tasks)


def test_global_fuzz_param_sets():
    rng = np.random.default_rng(105)
    for p in kswgen.fuzz_param_sets(rng, 25):
        pool, tasks, _ = kswgen.gen_glb_fuzz(rng, 150)
        _check_glb(p, pool, tasks)
