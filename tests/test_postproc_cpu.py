"""Region post-processing is host code: the library's own bmh_sort_and_dedup / bmh_mark_primary_se /
bmh_approx_mapq_se / bmh_pestat / bmh_pair are compared here, on the CPU, with outputs of the COMPILED REFERENCE's
mem_sort_and_dedup / mem_mark_primary_se / mem_approx_mapq_se / mem_pestat / mem_pair
 * committed in tests/golden/postproc_golden.npz (tools/make_postproc_fixture.py; vectors full of ties, because the
   reference's unstable introsort decides which duplicate survives and which hit is primary), and
 * live against oracle/_ref/libbwa_ref.so on further seeds, where that library exists (build container).
No GPU is involved; the SAM text itself is pinned end to end by tests/test_00_sam_parity.py."""
import ctypes as C
import os

import numpy as np
import pytest

import kswlib
import postgen
from __graft_entry__ import load_package

SAM_OPT = np.dtype([("a", "<i4"), ("b", "<i4"), ("o_del", "<i4"), ("e_del", "<i4"), ("o_ins", "<i4"), ("e_ins", "<i4"),
                    ("pen_unpaired", "<i4"), ("w", "<i4"), ("T", "<i4"), ("flag", "<i4"), ("min_seed_len", "<i4"), ("max_ins", "<i4"),
                    ("mapQ_coef_fac", "<i4"), ("max_matesw", "<i4"), ("mask_level", "<f4"), ("mask_level_redun", "<f4"),
                    ("mapQ_coef_len", "<f4"), ("mat", "i1", (25,)), ("pad", "i1", (3,))])
# mem_opt_init, reference bwamem.c:45-75
DEFAULTS = dict(a=1, b=4, o_del=6, e_del=1, o_ins=6, e_ins=1, pen_unpaired=17, w=100, T=30, flag=0, min_seed_len=19, max_ins=10000,
                mapQ_coef_fac=3, max_matesw=100, mask_level=0.5, mask_level_redun=0.95, mapQ_coef_len=50.0)


def sam_opt(**kw):
    o = np.zeros((), dtype=SAM_OPT)
    for k, v in {**DEFAULTS, **kw}.items():
        o[k] = v
    o["mapQ_coef_fac"] = int(np.log(float(o["mapQ_coef_len"]))) if float(o["mapQ_coef_len"]) > 0 else o["mapQ_coef_fac"]
    o["mat"] = kswlib.fill_scmat(int(o["a"]), int(o["b"]))
    return o


@pytest.fixture(scope="module")
def L():
    lib = load_package().lib()
    lib.bmh_sort_and_dedup.restype = C.c_int
    lib.bmh_sort_and_dedup.argtypes = [C.c_int, C.c_void_p, C.c_float]
    lib.bmh_mark_primary_se.restype = None
    lib.bmh_mark_primary_se.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
    lib.bmh_approx_mapq_se.restype = C.c_int
    lib.bmh_approx_mapq_se.argtypes = [C.c_void_p, C.c_void_p]
    lib.bmh_pestat.restype = None
    lib.bmh_pestat.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    lib.bmh_pair.restype = C.c_int
    return lib


def ours(L, o, vecs, ids0=12345):
    op = o.ctypes.data_as(C.c_void_p)
    ded, marked, mapq = [], [], []
    for i, v in enumerate(vecs):
        a = v.copy()
        n = L.bmh_sort_and_dedup(len(a), a.ctypes.data_as(C.c_void_p), C.c_float(float(o["mask_level_redun"])))
        a = a[:n].copy()
        ded.append(a.copy())
        L.bmh_mark_primary_se(op, len(a), a.ctypes.data_as(C.c_void_p), C.c_int64(ids0 + 7 * i))
        marked.append(a)
        mapq.append(np.array([L.bmh_approx_mapq_se(op, a[k:k + 1].ctypes.data_as(C.c_void_p)) for k in range(len(a))], dtype=np.int32))
    return ded, marked, mapq


def ours_pairs(L, o, pairs, l_pac):
    c_regs = kswlib.regs_to_c(pairs)
    pes = np.zeros(4, dtype=kswlib.PESTAT)
    L.bmh_pestat(o.ctypes.data_as(C.c_void_p), C.c_int64(l_pac), len(pairs), c_regs, pes.ctypes.data_as(C.c_void_p), -1)
    pr = np.zeros((len(pairs) // 2, 5), dtype=np.int32)
    for k in range(len(pairs) // 2):
        sub, nsub = C.c_int(0), C.c_int(0)
        z = (C.c_int * 2)(-1, -1)
        oo = L.bmh_pair(o.ctypes.data_as(C.c_void_p), C.c_int64(l_pac), pes.ctypes.data_as(C.c_void_p), C.byref(c_regs, 2 * k * C.sizeof(kswlib.CAlnregV)),
                        C.c_uint64(1000 + k), C.byref(sub), C.byref(nsub), z)
        pr[k] = (oo, sub.value, nsub.value, z[0], z[1])
    kswlib.regs_from_c(c_regs)
    return pes, pr


def split(flat, offs):
    return [flat[int(offs[i]): int(offs[i + 1])].copy() for i in range(len(offs) - 1)]


def test_postprocessing_matches_reference_fixture(L):
    g = np.load(os.path.join(kswlib.GOLDEN_DIR, "postproc_golden.npz"))
    n_ties = 0
    for si, kw in enumerate(postgen.OPTION_SETS):
        p = f"s{si}_"
        o = sam_opt(**kw)
        vecs = split(g[p + "in"], g[p + "in_off"])
        ded, marked, mapq = ours(L, o, vecs)
        want_ded = split(g[p + "ded"], g[p + "ded_off"])
        for i, (a, b) in enumerate(zip(ded, want_ded)):
            assert len(a) == len(b) and (a == b).all(), f"set {si} read {i}: dedup differs\nours={a}\nref={b}"
        assert (np.concatenate(marked) == g[p + "marked"]).all(), f"set {si}: primary marking differs"
        assert (np.concatenate(mapq) == g[p + "mapq"]).all(), f"set {si}: mapQ differs"
        n_ties += sum(int(len(v) - len(np.unique(v["re"]))) for v in vecs)
        pairs = split(g[p + "pairs"], g[p + "pairs_off"])
        pes, pr = ours_pairs(L, o, pairs, 1_000_000)
        for f in ("low", "high", "failed", "avg", "std"):
            assert (pes[f] == g[p + "pes"][f]).all(), (f, pes, g[p + "pes"])
        assert (pr == g[p + "pair_res"]).all(), f"set {si}: pairing differs at {np.nonzero((pr != g[p + 'pair_res']).any(axis=1))[0][:5]}"
        assert (pr[:, 0] > 0).sum() > 60
    assert n_ties > 250  # the tie-breaking of the unstable sorts is what this fixture is about


@pytest.mark.ref
def test_postprocessing_matches_live_reference(L):
    import reflib
    if not reflib.have_ref_bwa():
        pytest.skip("oracle/_ref not built")
    R = reflib.lib()
    R.mem_sort_and_dedup.restype = C.c_int
    R.mem_sort_and_dedup.argtypes = [C.c_int, C.c_void_p, C.c_float]
    R.mem_mark_primary_se.restype = None
    R.mem_mark_primary_se.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
    R.mem_approx_mapq_se.restype = C.c_int
    R.mem_approx_mapq_se.argtypes = [C.c_void_p, C.c_void_p]
    for seed in range(3):
        rng = np.random.default_rng(77 + seed)
        kw = postgen.OPTION_SETS[seed % len(postgen.OPTION_SETS)]
        opt = R.mem_opt_init()
        for k, v in kw.items():
            setattr(opt.contents, k, v)
        o = sam_opt(**kw)
        vecs = postgen.region_vectors(rng, 1500, 3_000_000)
        ded, marked, mapq = ours(L, o, vecs, ids0=99)
        for i, v in enumerate(vecs):
            a = v.copy()
            n = R.mem_sort_and_dedup(len(a), a.ctypes.data_as(C.c_void_p), C.c_float(opt.contents.mask_level_redun)) if len(a) else 0
            a = a[:n].copy()
            assert len(a) == len(ded[i]) and (a == ded[i]).all()
            if n:
                R.mem_mark_primary_se(opt, n, a.ctypes.data_as(C.c_void_p), C.c_int64(99 + 7 * i))
            assert (a == marked[i]).all()
            for k in range(n):
                assert R.mem_approx_mapq_se(opt, a[k:k + 1].ctypes.data_as(C.c_void_p)) == mapq[i][k]
