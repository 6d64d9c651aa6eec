#!/usr/bin/env python3
"""Long fuzz of the ksw_align2 oracle against the REFERENCE's own ksw_align2 (oracle/_ref/libksw_ref.so), build
container only.  Random matrices / gap penalties (including o = 0, where the reference's lazy-F loop degenerates), every
xtra combination, byte and word mode, planted hits, second copies, N's.  tests/test_oracle_vs_ref.py runs a short
version of the same; this one is for 10^5..10^6 cases:   python tools/fuzz_sw_vs_ref.py <seed> <cases>
Round 1: seeds 11, 12, 13 x 200 000 cases -> 0 mismatches (276 inputs skipped as undefined in the reference)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import kswlib
orc = kswlib.load_oracle()
ref = kswlib.load_ref_ksw()
class KR(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("score","te","qe","score2","te2","tb","qb")]
ref.ksw_align2.restype = KR
orc.orc_align2.restype = KR
def mkmat(a, b, amb=-1):
    m = np.zeros(25, np.int8)
    for i in range(4):
        for j in range(4):
            m[i*5+j] = a if i == j else -b
    for i in range(5):
        m[i*5+4] = amb; m[20+i] = amb
    return m
XB, XSTOP, XSUBO, XSTART = 0x10000, 0x20000, 0x40000, 0x80000
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
bad = 0; und_n = 0
t0 = time.time()
for it in range(N):
    a = int(rng.choice([1, 1, 1, 2, 3])); b = int(rng.choice([4, 4, 1, 2, 6]))
    od, ed, oi, ei = [int(x) for x in rng.choice([[6,1,6,1],[6,1,6,1],[0,1,0,1],[2,1,3,2],[4,2,6,1],[1,1,1,1],[0,2,0,3],[5,3,2,1]])]
    mat = mkmat(a, b, int(rng.choice([-1, -1, 0, -2])))
    if rng.random() < 0.1:
        mat = rng.integers(-6, 7, 25).astype(np.int8)
        if mat.max() <= 0: mat[0] = 2
    qlen = int(rng.choice([rng.integers(1, 20), rng.integers(20, 160), rng.integers(100, 300)]))
    tlen = int(rng.choice([rng.integers(1, 30), rng.integers(30, 400), rng.integers(200, 900)]))
    t = rng.integers(0, 4, tlen, dtype=np.uint8)
    mode = rng.random()
    if mode < 0.7 and tlen > 5:
        # query = mutated slice of the target (maybe partial)
        st = int(rng.integers(0, tlen)); ln = min(qlen, tlen - st)
        core = t[st:st+ln].copy()
        er = rng.choice([0.0, 0.02, 0.1, 0.3])
        mu = rng.random(ln) < er
        core[mu] = (core[mu] + rng.integers(1, 4, mu.sum())) & 3
        if ln > 10 and rng.random() < 0.5:
            c = int(rng.integers(2, ln - 2)); g = int(rng.integers(1, 8))
            if rng.random() < 0.5: core = np.concatenate([core[:c], core[c+g:]])
            else: core = np.concatenate([core[:c], rng.integers(0, 4, g, dtype=np.uint8), core[c:]])
        pre = rng.integers(0, 4, int(rng.integers(0, max(1, qlen - len(core) + 1))), dtype=np.uint8)
        q = np.concatenate([pre, core])[:qlen]
        if len(q) < qlen: q = np.concatenate([q, rng.integers(0, 4, qlen - len(q), dtype=np.uint8)])
        if rng.random() < 0.2:  # second copy of the hit elsewhere (score2)
            k = int(rng.integers(0, tlen)); l2 = min(len(core), tlen - k)
            t[k:k+l2] = core[:l2]
    else:
        q = rng.integers(0, 4, qlen, dtype=np.uint8)
    if rng.random() < 0.15:
        q[rng.random(qlen) < 0.05] = 4
    if rng.random() < 0.1:
        t[rng.random(tlen) < 0.03] = 4
    q = np.ascontiguousarray(q[:qlen]); qlen = len(q)
    xm = rng.random()
    thr = int(rng.choice([0, 10, 19, 19 * a, 30, 60]))
    if xm < 0.6: xtra = XSUBO | XSTART | thr
    elif xm < 0.7: xtra = XSTART
    elif xm < 0.8: xtra = XSUBO | thr
    elif xm < 0.9: xtra = XSTOP | thr
    else: xtra = 0
    shift = -int(mat.min()) if mat.min() < 0 else 0
    if rng.random() < 0.6:
        if qlen * int(mat.max()) < 250 - shift or not (xtra & XSTART) or rng.random() < 0.02: xtra |= XB
    und = C.c_int(0)
    q1, t1 = q.copy(), t.copy()
    r = ref.ksw_align2(qlen, q1.ctypes.data_as(C.c_void_p), tlen, t1.ctypes.data_as(C.c_void_p), 5, mat.ctypes.data_as(C.c_void_p), od, ed, oi, ei, xtra, None) if True else None
    o = orc.orc_align2(qlen, q.ctypes.data_as(C.c_void_p), tlen, t.ctypes.data_as(C.c_void_p), 5, mat.ctypes.data_as(C.c_void_p), od, ed, oi, ei, xtra, C.byref(und), None)
    if und.value:
        und_n += 1; continue
    rv = tuple(getattr(r, f) for f, _ in KR._fields_); ov = tuple(getattr(o, f) for f, _ in KR._fields_)
    if rv != ov:
        bad += 1
        if bad < 10: print("MISMATCH", it, qlen, tlen, hex(xtra), (a, b, od, ed, oi, ei), "ref", rv, "orc", ov)
print("done", N, "bad", bad, "undefined", und_n, "%.1fs" % (time.time() - t0))
