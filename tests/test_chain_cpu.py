"""Seeds to chains (bmh_chain_reads) is host code: compared here, on the CPU, with the chains the
COMPILED REFERENCE's mem_chain + mem_chain_flt produced for the same reads (tests/golden/chain_golden.npz,
tools/make_chain_fixture.py): same chains, same seeds, same ORDER -- on a repeat-rich genome where a read has up to
dozens of chains, so that the B-tree of chains splits and equal keys / equal weights occur."""
import ctypes as C
import os

import numpy as np
import pytest

import kswlib
from __graft_entry__ import load_package

CHAIN_OPT = np.dtype([("w", "<i4"), ("max_chain_gap", "<i4"), ("min_seed_len", "<i4"), ("max_occ", "<i4"), ("split_len", "<i4"),
                      ("split_width", "<i4"), ("mask_level", "<f4"), ("chain_drop_ratio", "<f4")])


class _Read(C.Structure):
    _fields_ = [("l_seq", C.c_int32), ("seq", C.c_void_p)]


class _Chain(C.Structure):
    _fields_ = [("n", C.c_int32), ("m", C.c_int32), ("pos", C.c_int64), ("seeds", C.c_void_p)]


class _ChainV(C.Structure):
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("a", C.POINTER(_Chain))]


def run_chain_reads(lib, o, l_pac, reads, calls, intvs, sa_k, sa_pos):
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    n = len(reads)
    c_reads = (_Read * n)()
    for k, r in enumerate(reads):
        c_reads[k].l_seq, c_reads[k].seq = len(r), r.ctypes.data
    call_off = np.concatenate([[0], np.cumsum([len(c) for c in calls])]).astype(np.uint32)
    intv_off = np.concatenate([[0], np.cumsum([len(v) for v in intvs])]).astype(np.uint64)
    fc = np.ascontiguousarray(np.concatenate(calls)) if n else np.zeros(0, kswlib.SMEM_CALL)
    fi = np.ascontiguousarray(np.concatenate(intvs)) if n else np.zeros(0, kswlib.SMEM_INTV)
    out = (_ChainV * n)()
    # the suffix-array positions in interval order, as the library's own key list asks for them (one bmh_sa_batch on the
    # GPU in production; here they come out of the fixture's sorted table)
    lib.bmh_chain_sa_keys.restype = C.c_uint64
    sa_off = np.zeros(len(fi) + 1, dtype=np.uint64)
    nk = lib.bmh_chain_sa_keys(o.ctypes.data_as(C.c_void_p), C.c_uint64(len(fi)), fi.ctypes.data_as(C.c_void_p), sa_off.ctypes.data_as(C.c_void_p), None)
    keys = np.zeros(nk + 1, dtype=np.uint64)
    assert lib.bmh_chain_sa_keys(o.ctypes.data_as(C.c_void_p), C.c_uint64(len(fi)), fi.ctypes.data_as(C.c_void_p), sa_off.ctypes.data_as(C.c_void_p),
                                 keys.ctypes.data_as(C.c_void_p)) == nk
    at = np.searchsorted(sa_k, keys[:nk])
    assert (sa_k[at] == keys[:nk]).all()
    pos = np.ascontiguousarray(np.concatenate([sa_pos[at], np.zeros(1, np.uint64)]))
    rc = lib.bmh_chain_reads(o.ctypes.data_as(C.c_void_p), C.c_int64(l_pac), C.c_int(n), C.cast(c_reads, C.c_void_p),
                             call_off.ctypes.data_as(C.c_void_p), fc.ctypes.data_as(C.c_void_p), intv_off.ctypes.data_as(C.c_void_p),
                             fi.ctypes.data_as(C.c_void_p), sa_off.ctypes.data_as(C.c_void_p), pos.ctypes.data_as(C.c_void_p),
                             C.cast(out, C.c_void_p))
    assert rc == 0, rc
    res = []
    for k in range(n):
        chains = []
        for ci in range(out[k].n):
            c = out[k].a[ci]
            sd = np.zeros(c.n, dtype=kswlib.SEED)
            C.memmove(sd.ctypes.data, c.seeds, c.n * kswlib.SEED.itemsize)
            chains.append(sd)
            libc.free(c.seeds)
        if out[k].a:
            libc.free(C.cast(out[k].a, C.c_void_p))
        res.append(chains)
    return res


def test_chains_match_reference_fixture():
    lib = load_package().lib()
    lib.bmh_chain_reads.restype = C.c_int
    g = np.load(os.path.join(kswlib.GOLDEN_DIR, "chain_golden.npz"))
    l_pac = int(g["l_pac"])
    total, deep, ties = 0, 0, 0
    for p in [str(x) for x in g["groups"]]:
        o = np.zeros((), dtype=CHAIN_OPT)
        for f, v in zip(CHAIN_OPT.names[:6], g[p + "opt"]):
            o[f] = v
        o["mask_level"], o["chain_drop_ratio"] = g[p + "optf"]
        cut = lambda flat, cnt: np.split(flat, np.cumsum(cnt)[:-1])
        reads = [np.ascontiguousarray(r) for r in cut(g[p + "reads"], g[p + "read_len"])]
        calls, intvs = cut(g[p + "calls"], g[p + "n_calls"]), cut(g[p + "intv"], g[p + "n_intv"])
        got = run_chain_reads(lib, o, l_pac, reads, calls, intvs, np.ascontiguousarray(g[p + "sa_k"]), np.ascontiguousarray(g[p + "sa_pos"]))
        nseeds = cut(g[p + "seeds"], g[p + "n_seeds"]) if len(g[p + "n_seeds"]) else []
        it = iter(nseeds)
        for r, nch in enumerate(g[p + "n_chains"]):
            want = [next(it) for _ in range(int(nch))]
            assert len(got[r]) == len(want), f"{p} read {r}: {len(got[r])} chains, reference {len(want)}"
            for ci, (a, b) in enumerate(zip(got[r], want)):
                assert len(a) == len(b) and (a == b).all(), f"{p} read {r} chain {ci}: ours={a} ref={b}"
            total += len(want)
            deep += len(want) > 15
            pos = [int(c["rbeg"][0]) for c in want]
            ties += len(pos) - len(set(pos))
    assert total > 3000 and deep > 30  # reads whose chains split the B-tree's root (more than 2t-1 = 15 keys)
