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


def _check_sw(p, pool, tasks):
    ref = kswlib.ref_sw_batch(p, pool, tasks)
    orc, _ = kswlib.orc_sw_batch(p, pool, tasks, nthreads=4)
    assert (orc["rsv"] == 0).all()
    for f in kswlib.SW_FIELDS:
        bad = np.nonzero(ref[f] != orc[f])[0]
        assert len(bad) == 0, f"{f}: first mismatch task {tasks[bad[0]]} ref={ref[bad[0]]} orc={orc[bad[0]]}"


def test_sw_materescue_shapes():
    """ksw_align2 as mem_matesw calls it (reference bwamem_pair.c:147-148): byte mode, KSW_XSUBO|KSW_XSTART|19."""
    rng = np.random.default_rng(111)
    p = kswlib.make_params()
    pool, tasks = kswgen.gen_sw_materescue(rng, 600, p)
    _check_sw(p, pool, tasks)
    pool, tasks = kswgen.gen_sw_materescue(rng, 500, p, read_len=(60, 280), win=(40, 600), hard=True)
    _check_sw(p, pool, tasks)


def test_sw_fuzz_param_sets():
    """All xtra combinations, byte and word mode, tiny sizes, random matrices; includes o_ins = 0, where the
    reference's lazy-F loop exits after one column (the oracle keeps that loop literally)."""
    rng = np.random.default_rng(112)
    sets = kswgen.sw_param_sets(rng, 14) + [kswlib.make_params(o_del=0, o_ins=0), kswlib.make_params(o_del=3, o_ins=0, e_ins=2)]
    for p in sets:
        pool, tasks = kswgen.gen_sw_fuzz(rng, 300, p)
        _check_sw(p, pool, tasks)
