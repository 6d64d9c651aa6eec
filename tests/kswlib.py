"""Shared test plumbing: record layouts, ctypes bindings of the CHECKERS.

* oracle/liborc.so         -- our CPU restatement (always available; built on demand)
* oracle/_ref/libksw_ref.so, libbwa_ref.so -- the reference compiled from
  /root/reference by oracle/Makefile (only where that tree exists or the
  prebuilt files travelled with the repo).

Nothing in here is product code; the product is libbwamem_hip.so, bound in
bwa-mem-quickassist_amd/__init__.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

# ---- record layouts == include/bwamem_hip.h -------------------------------
EXT_TASK = np.dtype([("q_off", "<u8"), ("t_off", "<u8"), ("qlen", "<u2"), ("tlen", "<u2"),
                     ("h0", "<i4"), ("w", "<i2"), ("end_bonus", "<i2"), ("flags", "<u2"),
                     ("rsv", "<u2")])
EXT_RES = np.dtype([("score", "<i4"), ("qle", "<i4"), ("tle", "<i4"), ("gtle", "<i4"),
                    ("gscore", "<i4"), ("max_off", "<i4")])
GLB_TASK = np.dtype([("q_off", "<u8"), ("t_off", "<u8"), ("qlen", "<u2"), ("tlen", "<u2"),
                     ("w", "<i4"), ("cigar_off", "<u4"), ("cigar_cap", "<u4")])
GLB_RES = np.dtype([("score", "<i4"), ("n_cigar", "<i4")])
SEED_TASK = np.dtype([("q_off", "<u8"), ("t_off", "<u8"), ("l_query", "<i4"), ("qbeg", "<i4"), ("len", "<i4"),
                      ("rbeg", "<i4"), ("wlen", "<i4"), ("flags", "<u2"), ("rsv", "<u2")])
SEED_RES = np.dtype([("qb", "<i4"), ("qe", "<i4"), ("rb", "<i4"), ("re", "<i4"), ("score", "<i4"), ("truesc", "<i4"),
                     ("w", "<i4"), ("n_ext", "<i4")])
assert SEED_TASK.itemsize == 40 and SEED_RES.itemsize == 32
PARAMS = np.dtype([("o_del", "<i4"), ("e_del", "<i4"), ("o_ins", "<i4"), ("e_ins", "<i4"),
                   ("zdrop", "<i4"), ("a", "<i4"), ("w", "<i4"), ("pen_clip5", "<i4"),
                   ("pen_clip3", "<i4"), ("mat", "i1", (25,)), ("pad", "i1", (3,))])
SEED = np.dtype([("rbeg", "<i8"), ("qbeg", "<i4"), ("len", "<i4")])
ALNREG = np.dtype([("rb", "<i8"), ("re", "<i8"), ("qb", "<i4"), ("qe", "<i4"), ("score", "<i4"),
                   ("truesc", "<i4"), ("sub", "<i4"), ("csub", "<i4"), ("sub_n", "<i4"),
                   ("w", "<i4"), ("seedcov", "<i4"), ("secondary", "<i4"), ("hash", "<u8")])
SW_TASK = np.dtype([("q_off", "<u8"), ("t_off", "<u8"), ("tlen", "<u4"), ("qlen", "<u2"), ("flags", "<u2"),
                    ("xtra", "<u4"), ("rsv", "<u4")])
SW_RES = np.dtype([("score", "<i4"), ("te", "<i4"), ("qe", "<i4"), ("score2", "<i4"), ("te2", "<i4"),
                   ("tb", "<i4"), ("qb", "<i4"), ("rsv", "<i4")])
KSW_XBYTE, KSW_XSTOP, KSW_XSUBO, KSW_XSTART = 0x10000, 0x20000, 0x40000, 0x80000  # reference ksw.h:6-9
PESTAT = np.dtype([("low", "<i4"), ("high", "<i4"), ("failed", "<i4"), ("pad", "<i4"), ("avg", "<f8"), ("std", "<f8")])
MATESW_OPT = np.dtype([("pen_unpaired", "<i4"), ("max_matesw", "<i4"), ("min_seed_len", "<i4"), ("rsv", "<i4")])
assert PESTAT.itemsize == 32
SW_FIELDS = ("score", "te", "qe", "score2", "te2", "tb", "qb")
assert SW_TASK.itemsize == 32 and SW_RES.itemsize == 32
assert EXT_TASK.itemsize == 32 and EXT_RES.itemsize == 24
assert GLB_TASK.itemsize == 32 and GLB_RES.itemsize == 8
assert PARAMS.itemsize == 64 and SEED.itemsize == 16 and ALNREG.itemsize == 64

BMH_F_QREV, BMH_F_TREV, BMH_F_TPAC, BMH_F_QCOMP = 1, 2, 4, 8


def fill_scmat(a, b, n_score=-1):
    """bwa_fill_scmat, reference bwa.c:77-86: diag a, off-diag -b, row/col 4 = -1."""
    m = np.full((5, 5), -b, dtype=np.int8)
    for i in range(4):
        m[i, i] = a
    m[4, :] = n_score
    m[:, 4] = n_score
    return m.reshape(25)


def make_params(a=1, b=4, o_del=6, e_del=1, o_ins=6, e_ins=1, w=100, zdrop=100,
                pen_clip5=5, pen_clip3=5, mat=None):
    """Defaults = mem_opt_init, reference bwamem.c:45-75."""
    p = np.zeros((), dtype=PARAMS)
    p["a"], p["o_del"], p["e_del"], p["o_ins"], p["e_ins"] = a, o_del, e_del, o_ins, e_ins
    p["w"], p["zdrop"], p["pen_clip5"], p["pen_clip3"] = w, zdrop, pen_clip5, pen_clip3
    p["mat"] = fill_scmat(a, b) if mat is None else np.asarray(mat, dtype=np.int8)
    return p


# ---- ctypes mirrors ---------------------------------------------------------
class OrcScoring(C.Structure):
    _fields_ = [("o_del", C.c_int), ("e_del", C.c_int), ("o_ins", C.c_int), ("e_ins", C.c_int),
                ("zdrop", C.c_int), ("m", C.c_int), ("mat", C.POINTER(C.c_int8))]


class Kswr(C.Structure):  # kswr_t, reference ksw.h:14-19
    _fields_ = [(n, C.c_int) for n in ("score", "te", "qe", "score2", "te2", "tb", "qb")]


class OrcExtOut(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("score", "qle", "tle", "gtle", "gscore", "max_off")]


def _make(target=None):
    cmd = ["make", "-s", "-C", ORACLE_DIR] + ([target] if target else [])
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)


_orc = None


def load_oracle():
    global _orc
    if _orc is None:
        path = os.path.join(ORACLE_DIR, "liborc.so")
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
        srcs.append(os.path.join(ROOT, "include", "bwamem_hip.h"))
        if not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs):
            _make("liborc.so")
        lib = C.CDLL(path)
        lib.orc_extend.restype = None
        lib.orc_global.restype = C.c_int
        lib.orc_extend_batch.restype = C.c_int
        lib.orc_extend_batch_pac.restype = C.c_int
        lib.orc_align2.restype = Kswr
        lib.orc_sw_batch.restype = C.c_int
        lib.orc_chain2aln.restype = None
        lib.orc_get_seq.restype = C.c_void_p
        lib.orc_cal_max_gap.restype = C.c_int
        _orc = lib
    return _orc


def have_ref():
    return os.path.exists(os.path.join(REF_DIR, "libksw_ref.so"))


_ref_ksw = None


def load_ref_ksw():
    """The reference's own ksw.c compiled by oracle/Makefile (None if absent)."""
    global _ref_ksw
    if _ref_ksw is None and have_ref():
        lib = C.CDLL(os.path.join(REF_DIR, "libksw_ref.so"))
        lib.ksw_extend2.restype = C.c_int
        lib.ksw_global2.restype = C.c_int
        lib.ksw_align2.restype = Kswr
        _ref_ksw = lib
    return _ref_ksw


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def scoring_of(p, keep):
    mat = np.ascontiguousarray(p["mat"], dtype=np.int8)
    keep.append(mat)
    return OrcScoring(int(p["o_del"]), int(p["e_del"]), int(p["o_ins"]), int(p["e_ins"]),
                      int(p["zdrop"]), 5, mat.ctypes.data_as(C.POINTER(C.c_int8)))


def task_seqs(pool, t):
    """Materialise (query, target) of one EXT_TASK/GLB_TASK honouring the REV flags."""
    def get(off, n, rev):
        off, n = int(off), int(n)
        if rev:
            return np.ascontiguousarray(pool[off - n + 1: off + 1][::-1]) if n else np.zeros(0, np.uint8)
        return np.ascontiguousarray(pool[off: off + n])
    flags = int(t["flags"]) if "flags" in t.dtype.names else 0
    return get(t["q_off"], t["qlen"], flags & BMH_F_QREV), get(t["t_off"], t["tlen"], flags & BMH_F_TREV)


def ref_extend_batch(p, pool, tasks):
    """Run the compiled REFERENCE ksw_extend2 on every task."""
    lib = load_ref_ksw()
    out = np.zeros(len(tasks), dtype=EXT_RES)
    mat = np.ascontiguousarray(p["mat"], dtype=np.int8)
    ints = [C.c_int() for _ in range(5)]
    for k, t in enumerate(tasks):
        q, tg = task_seqs(pool, t)
        sc = lib.ksw_extend2(int(t["qlen"]), _u8p(q), int(t["tlen"]), _u8p(tg), 5,
                             mat.ctypes.data_as(C.POINTER(C.c_int8)), int(p["o_del"]), int(p["e_del"]),
                             int(p["o_ins"]), int(p["e_ins"]), int(t["w"]), int(t["end_bonus"]),
                             int(p["zdrop"]), int(t["h0"]), *[C.byref(x) for x in ints])
        out[k] = (sc,) + tuple(x.value for x in ints)
    return out


def orc_extend_batch(p, pool, tasks, nthreads=1, pac=None, l_pac=0):
    """Run OUR restatement on every task; returns (results, total_cells).  `pac`/`l_pac` serve BMH_F_TPAC tasks."""
    lib = load_oracle()
    if pac is not None:
        pac = np.ascontiguousarray(pac, dtype=np.uint8)
    keep = []
    sc = scoring_of(p, keep)
    tasks = np.ascontiguousarray(tasks)
    pool = np.ascontiguousarray(pool)
    out = np.zeros(len(tasks), dtype=EXT_RES)
    cells = C.c_int64(0)
    lib.orc_extend_batch_pac(C.byref(sc), pool.ctypes.data_as(C.c_void_p),
                             pac.ctypes.data_as(C.c_void_p) if pac is not None else None, C.c_int64(l_pac),
                             tasks.ctypes.data_as(C.c_void_p), C.c_int(len(tasks)), out.ctypes.data_as(C.c_void_p),
                             C.byref(cells), C.c_int(nthreads))
    return out, cells.value


def orc_seedext_batch(p, pool, tasks, nthreads=1, pac=None, l_pac=0):
    """OUR restatement of bwamem.c:808-866 per seed record; returns (results SEED_RES[], cells, ksw_extend2 calls)."""
    lib = load_oracle()
    pp = np.ascontiguousarray(np.asarray(p, dtype=PARAMS).reshape(()))
    tasks = np.ascontiguousarray(tasks, dtype=SEED_TASK)
    pool = np.ascontiguousarray(pool, dtype=np.uint8)
    if pac is not None:
        pac = np.ascontiguousarray(pac, dtype=np.uint8)
    out = np.zeros(len(tasks), dtype=SEED_RES)
    cells, calls = C.c_int64(0), C.c_int64(0)
    lib.orc_seedext_batch(pp.ctypes.data_as(C.c_void_p), pool.ctypes.data_as(C.c_void_p),
                          pac.ctypes.data_as(C.c_void_p) if pac is not None else None, C.c_int64(l_pac),
                          tasks.ctypes.data_as(C.c_void_p), C.c_int64(len(tasks)), out.ctypes.data_as(C.c_void_p),
                          C.byref(cells), C.byref(calls), C.c_int(nthreads))
    return out, cells.value, calls.value


def sw_task_seqs(pool, t, pac=None, l_pac=0):
    """Materialise (query, target) of one SW_TASK honouring QREV/QCOMP/TREV/TPAC."""
    flags, ql, tl = int(t["flags"]), int(t["qlen"]), int(t["tlen"])
    qo, to = int(t["q_off"]), int(t["t_off"])
    q = pool[qo - ql + 1: qo + 1][::-1] if flags & BMH_F_QREV else pool[qo: qo + ql]
    q = np.ascontiguousarray(q).copy()
    if flags & BMH_F_QCOMP:
        q = np.where(q < 4, 3 - q, 4).astype(np.uint8)
    if flags & BMH_F_TPAC:
        pos = to - np.arange(tl) if flags & BMH_F_TREV else to + np.arange(tl)
        f = np.where(pos >= l_pac, 2 * l_pac - 1 - pos, pos)
        b = (pac[f >> 2] >> ((~f & 3) << 1)) & 3
        tg = np.where(pos >= l_pac, 3 - b, b).astype(np.uint8)
    else:
        tg = pool[to - tl + 1: to + 1][::-1] if flags & BMH_F_TREV else pool[to: to + tl]
    return q, np.ascontiguousarray(tg).copy()


def ref_sw_batch(p, pool, tasks, pac=None, l_pac=0):
    """Run the compiled REFERENCE ksw_align2 (qry = NULL) on every task."""
    lib = load_ref_ksw()
    out = np.zeros(len(tasks), dtype=SW_RES)
    mat = np.ascontiguousarray(p["mat"], dtype=np.int8)
    for k, t in enumerate(tasks):
        q, tg = sw_task_seqs(pool, t, pac, l_pac)
        r = lib.ksw_align2(len(q), _u8p(q), len(tg), _u8p(tg), 5, mat.ctypes.data_as(C.POINTER(C.c_int8)),
                           int(p["o_del"]), int(p["e_del"]), int(p["o_ins"]), int(p["e_ins"]), int(t["xtra"]), None)
        out[k] = tuple(getattr(r, f) for f in SW_FIELDS) + (0,)
    return out


def ref_sw_batch_mt(p, pool, tasks, nthreads=1):
    """The compiled REFERENCE ksw_align2 over all tasks on `nthreads` host threads (oracle/ref_batch_shim.c)."""
    lib = load_ref_ksw()
    tasks = np.ascontiguousarray(tasks, dtype=SW_TASK)
    pool = np.ascontiguousarray(pool, dtype=np.uint8)
    pp = np.ascontiguousarray(p, dtype=PARAMS)
    out = np.zeros(len(tasks), dtype=SW_RES)
    lib.ref_sw_batch_mt(pp.ctypes.data_as(C.c_void_p), pool.ctypes.data_as(C.c_void_p), tasks.ctypes.data_as(C.c_void_p),
                        C.c_int(len(tasks)), out.ctypes.data_as(C.c_void_p), C.c_int(nthreads))
    return out


def orc_sw_batch(p, pool, tasks, nthreads=1, pac=None, l_pac=0):
    """OUR restatement of ksw_align2 on every task; returns (results, cells).  results["rsv"] = 1 marks inputs for
    which the reference's behaviour is undefined (byte overflow + KSW_XSTART)."""
    lib = load_oracle()
    tasks = np.ascontiguousarray(tasks, dtype=SW_TASK)
    pool = np.ascontiguousarray(pool, dtype=np.uint8)
    pp = np.ascontiguousarray(p, dtype=PARAMS)
    if pac is not None:
        pac = np.ascontiguousarray(pac, dtype=np.uint8)
    out = np.zeros(len(tasks), dtype=SW_RES)
    cells = C.c_int64(0)
    lib.orc_sw_batch(pp.ctypes.data_as(C.c_void_p), pool.ctypes.data_as(C.c_void_p),
                     pac.ctypes.data_as(C.c_void_p) if pac is not None else None, C.c_int64(l_pac),
                     tasks.ctypes.data_as(C.c_void_p), C.c_int(len(tasks)), out.ctypes.data_as(C.c_void_p),
                     C.byref(cells), C.c_int(nthreads))
    return out, cells.value


def _global_generic(fn_call, pool, tasks):
    res = np.zeros(len(tasks), dtype=GLB_RES)
    cigars = []
    for k, t in enumerate(tasks):
        q, tg = task_seqs(pool, t)
        n = C.c_int(0)
        cg = C.POINTER(C.c_uint32)()
        want = int(t["cigar_cap"]) > 0
        sc = fn_call(int(t["qlen"]), _u8p(q), int(t["tlen"]), _u8p(tg), int(t["w"]),
                     C.byref(n) if want else None, C.byref(cg) if want else None)
        arr = np.array([cg[i] for i in range(n.value)], dtype=np.uint32)
        if want and n.value:
            _libc.free(cg)
        res[k] = (sc, n.value)
        cigars.append(arr)
    return res, cigars


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def ref_global_batch(p, pool, tasks):
    lib = load_ref_ksw()
    mat = np.ascontiguousarray(p["mat"], dtype=np.int8)

    def call(ql, q, tl, t, w, n, cg):
        return lib.ksw_global2(ql, q, tl, t, 5, mat.ctypes.data_as(C.POINTER(C.c_int8)), int(p["o_del"]),
                               int(p["e_del"]), int(p["o_ins"]), int(p["e_ins"]), w, n, cg)
    return _global_generic(call, pool, tasks)


def orc_global_batch(p, pool, tasks):
    lib = load_oracle()
    keep = []
    sc = scoring_of(p, keep)

    def call(ql, q, tl, t, w, n, cg):
        return lib.orc_global(C.byref(sc), ql, q, tl, t, w, n, cg)
    return _global_generic(call, pool, tasks)


# ---- driver level ------------------------------------------------------------
class _CSeedChain(C.Structure):  # bmh_chain_t == mem_chain_t
    _fields_ = [("n", C.c_int32), ("m", C.c_int32), ("pos", C.c_int64), ("seeds", C.c_void_p)]


class _CAlnregV(C.Structure):
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("a", C.c_void_p)]


def orc_chain2aln_reads(p, l_pac, pac, reads, chains):
    """OUR restatement of mem_chain2aln, read by read, chain by chain (shared region vector per read)."""
    lib = load_oracle()
    pp = np.ascontiguousarray(np.asarray(p, dtype=PARAMS).reshape(()))
    pac = np.ascontiguousarray(pac, dtype=np.uint8)
    out = []
    for seq, chs in zip(reads, chains):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        regs = _CAlnregV(0, 0, None)
        for sd in chs:
            sd = np.ascontiguousarray(sd, dtype=SEED)
            c = _CSeedChain(len(sd), len(sd), int(sd["rbeg"][0]) if len(sd) else 0, sd.ctypes.data)
            lib.orc_chain2aln(pp.ctypes.data_as(C.c_void_p), C.c_int64(l_pac), pac.ctypes.data_as(C.c_void_p),
                              C.c_int(len(seq)), seq.ctypes.data_as(C.c_void_p), C.byref(c), C.byref(regs), None)
        a = np.zeros(regs.n, dtype=ALNREG)
        if regs.n:
            C.memmove(a.ctypes.data, regs.a, regs.n * ALNREG.itemsize)
        if regs.a:
            _libc.free(regs.a)
        out.append(a)
    return out


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=False)


def golden_chain2aln_groups():
    """Yields (params, l_pac, pac, reads, chains, expected_regs) per parameter set of chain2aln_golden.npz."""
    g = load_golden("chain2aln_golden.npz")
    l_pac, pac = int(g["l_pac"]), g["pac"]
    ro, rp = g["read_off"], g["read_pool"]
    nch, nsd, seeds = g["read_nchains"], g["chain_nseeds"], g["seeds"]
    nrg, regs = g["read_nregs"], g["regs"]
    ci = si = gi = 0
    per_group = {}
    for r in range(len(nch)):
        read = rp[ro[r]:ro[r + 1]]
        chs = []
        for _ in range(nch[r]):
            chs.append(seeds[si:si + nsd[ci]])
            si += nsd[ci]
            ci += 1
        exp = regs[gi:gi + nrg[r]]
        gi += nrg[r]
        per_group.setdefault(int(g["read_group"][r]), []).append((read, chs, exp))
    for k in sorted(per_group):
        items = per_group[k]
        yield g["params"][k], l_pac, pac, [x[0] for x in items], [x[1] for x in items], [x[2] for x in items]


# ---- phase 2: CIGAR generation (row a6) ---------------------------------------
CIGAR_REQ = np.dtype([("read", "<i4"), ("qb", "<i4"), ("qe", "<i4"), ("pad", "<i4"), ("rb", "<i8"), ("re", "<i8"),
                      ("truesc", "<i4"), ("reg_w", "<i4")])


class _OrcCigar(C.Structure):
    _fields_ = [("score", C.c_int), ("n_cigar", C.c_int), ("NM", C.c_int), ("w_used", C.c_int),
                ("cigar", C.POINTER(C.c_uint32)), ("md", C.c_char_p)]


def orc_reg2cigar(p, l_pac, pac, read, req):
    """OUR restatement of mem_reg2aln's band/retry loop over bwa_gen_cigar2: (score, cigar words, NM, MD, tries)."""
    lib = load_oracle()
    lib.orc_reg2cigar.restype = None
    pp = np.ascontiguousarray(np.asarray(p, dtype=PARAMS).reshape(()))
    pac = np.ascontiguousarray(pac, dtype=np.uint8)
    read = np.ascontiguousarray(read, dtype=np.uint8)
    o, rounds = _OrcCigar(), C.c_int(0)
    lib.orc_reg2cigar(pp.ctypes.data_as(C.c_void_p), C.c_int64(l_pac), pac.ctypes.data_as(C.c_void_p),
                      read.ctypes.data_as(C.c_void_p), C.c_int(int(req["qb"])), C.c_int(int(req["qe"])),
                      C.c_int64(int(req["rb"])), C.c_int64(int(req["re"])), C.c_int(int(req["truesc"])),
                      C.c_int(int(req["reg_w"])), C.byref(o), C.byref(rounds))
    words = np.array([o.cigar[i] for i in range(o.n_cigar)], dtype=np.uint32)
    md = bytes(o.md) if o.md else b""
    res = (o.score, words, o.NM, md, rounds.value)
    lib.orc_cigar_free(C.byref(o))
    return res


def finish_aln(words, md, req, l_query, l_pac):
    """What mem_reg2aln does AFTER the CIGAR loop (reference bwamem.c:1202-1230), so that the loop's output can
    be compared with the reference's mem_aln_t: drop a leading/trailing deletion, add soft clips."""
    words = [int(x) for x in words]
    is_rev = int(req["rb"]) >= l_pac
    if words:
        if words[0] & 0xf == 2:
            words = words[1:]
        elif words[-1] & 0xf == 2:
            words = words[:-1]
    qb, qe = int(req["qb"]), int(req["qe"])
    if qb != 0 or qe != l_query:
        clip5 = l_query - qe if is_rev else qb
        clip3 = qb if is_rev else l_query - qe
        if clip5:
            words = [clip5 << 4 | 3] + words
        if clip3:
            words = words + [clip3 << 4 | 3]
    return np.array(words, dtype=np.uint32), md


def golden_cigar_groups():
    """Yields (params, l_pac, pac, reads, reqs, expected) per parameter set of cigar_golden.npz;
    expected = list of (n_cigar, words, NM, md_bytes) from the reference's own mem_reg2aln."""
    g = load_golden("cigar_golden.npz")
    l_pac, pac = int(g["l_pac"]), g["pac"]
    ro, rp = g["read_off"], g["read_pool"]
    reads = [rp[ro[r]:ro[r + 1]] for r in range(len(ro) - 1)]
    mds = bytes(g["exp_md"]).split(b"\0")
    woff = np.concatenate([[0], np.cumsum(g["exp_n_cigar"])])
    for k in range(len(g["params"])):
        sel = np.nonzero(g["group"] == k)[0]
        exp = [(int(g["exp_n_cigar"][i]), g["exp_cigar"][woff[i]:woff[i + 1]], int(g["exp_nm"][i]), mds[i]) for i in sel]
        yield g["params"][k], l_pac, pac, reads, g["reqs"][sel], exp


def orc_global_batch_mt(p, pool, tasks, cigar_words, nthreads=1):
    """Threaded oracle run of N x ksw_global2 (for the CPU baseline): (results, cigar_pool, band_cells)."""
    lib = load_oracle()
    keep = []
    sc = scoring_of(p, keep)
    tasks = np.ascontiguousarray(tasks)
    pool = np.ascontiguousarray(pool)
    res = np.zeros(len(tasks), dtype=GLB_RES)
    cig = np.zeros(max(int(cigar_words), 1), dtype=np.uint32)
    cells = C.c_int64(0)
    lib.orc_global_batch(C.byref(sc), pool.ctypes.data_as(C.c_void_p), tasks.ctypes.data_as(C.c_void_p),
                         C.c_int(len(tasks)), res.ctypes.data_as(C.c_void_p), cig.ctypes.data_as(C.c_void_p),
                         C.byref(cells), C.c_int(nthreads))
    return res, cig, cells.value


# ---- mate rescue (mem_matesw loop) ------------------------------------------------------------------------------
class CRead(C.Structure):  # bmh_read_t
    _fields_ = [("l_seq", C.c_int32), ("seq", C.c_void_p)]


class CAlnregV(C.Structure):  # bmh_alnreg_v == mem_alnreg_v
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("a", C.c_void_p)]


DEDUP_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p)
_libc = C.CDLL(None)
_libc.malloc.restype = C.c_void_p
_libc.malloc.argtypes = [C.c_size_t]
_libc.free.argtypes = [C.c_void_p]


def regs_to_c(regs_list):
    """List of ALNREG arrays -> C array of bmh_alnreg_v whose .a are malloc()ed (the callee realloc()s them)."""
    arr = (CAlnregV * len(regs_list))()
    for k, r in enumerate(regs_list):
        r = np.ascontiguousarray(r, dtype=ALNREG)
        arr[k].n = arr[k].m = len(r)
        if len(r):
            arr[k].a = _libc.malloc(len(r) * ALNREG.itemsize)
            C.memmove(arr[k].a, r.ctypes.data, len(r) * ALNREG.itemsize)
    return arr


def regs_from_c(arr, free=True):
    out = []
    for k in range(len(arr)):
        r = np.zeros(arr[k].n, dtype=ALNREG)
        if arr[k].n:
            C.memmove(r.ctypes.data, arr[k].a, arr[k].n * ALNREG.itemsize)
        if free and arr[k].a:
            _libc.free(arr[k].a)
        out.append(r)
    return out


def reads_to_c(reads, keep):
    arr = (CRead * len(reads))()
    for k, r in enumerate(reads):
        r = np.ascontiguousarray(r, dtype=np.uint8)
        keep.append(r)
        arr[k].l_seq, arr[k].seq = len(r), r.ctypes.data
    return arr


def simple_dedup_fn():
    """orc_simple_dedup as a C function pointer (shared by DUT and oracle where the reference is absent)."""
    return C.cast(load_oracle().orc_simple_dedup, C.c_void_p)


def orc_matesw_pairs(p, o, l_pac, pac, pes, reads, regs, dedup):
    """OUR sequential restatement of the mate-rescue block of mem_sam_pe over every pair.
    reads/regs: flat lists, 2 per pair.  Returns (regs after rescue, n per pair)."""
    lib = load_oracle()
    lib.orc_matesw_pair.restype = C.c_int
    pp = np.ascontiguousarray(p, dtype=PARAMS)
    oo = np.ascontiguousarray(o, dtype=MATESW_OPT)
    pes = np.ascontiguousarray(pes, dtype=PESTAT)
    pac = np.ascontiguousarray(pac, dtype=np.uint8)
    keep = []
    c_reads = reads_to_c(reads, keep)
    c_regs = regs_to_c(regs)
    ns = []
    for k in range(len(reads) // 2):
        ns.append(lib.orc_matesw_pair(pp.ctypes.data_as(C.c_void_p), oo.ctypes.data_as(C.c_void_p), C.c_int64(l_pac),
                                      pac.ctypes.data_as(C.c_void_p), pes.ctypes.data_as(C.c_void_p),
                                      C.byref(c_reads, 2 * k * C.sizeof(CRead)), C.byref(c_regs, 2 * k * C.sizeof(CAlnregV)),
                                      dedup, None))
    return regs_from_c(c_regs), ns


def golden_matesw_groups():
    """Yields (params, opt, pes, l_pac, pac, reads, regs, expect_regs, n_sw) per group of matesw_golden.npz."""
    g = load_golden("matesw_golden.npz")
    l_pac, pac = int(g["l_pac"]), g["pac"]

    def split(flat, counts):
        out, o = [], 0
        for c in counts:
            out.append(flat[o:o + c].copy())
            o += c
        return out
    for key in g["groups"]:
        key = str(key)
        reads = split(g[key + "reads"], g[key + "read_len"])
        regs = split(g[key + "regs"], g[key + "regs_n"])
        exp = split(g[key + "exp"], g[key + "exp_n"])
        yield g[key + "params"], g[key + "opt"], g[key + "pes"], l_pac, pac, reads, regs, exp, g[key + "n_sw"].tolist()


# ---- FM-index (seeding) ---------------------------------------------------------------------------------------------
SMEM_INTV = np.dtype([("x0", "<u8"), ("x1", "<u8"), ("x2", "<u8"), ("info", "<u8")])
SMEM_CALL = np.dtype([("x", "<i4"), ("min_intv", "<i4"), ("ret", "<i4"), ("n", "<i4"), ("first", "<u4"), ("rsv", "<u4")])
SMEM_OPT = np.dtype([("min_seed_len", "<i4"), ("split_len", "<i4"), ("split_width", "<i4"), ("start_width", "<i4"), ("min_emit_len", "<i4")])


class CBwt(C.Structure):  # bmh_bwt_t
    _fields_ = [("primary", C.c_uint64), ("L2", C.c_uint64 * 5), ("seq_len", C.c_uint64), ("bwt_size", C.c_uint64),
                ("bwt", C.c_void_p), ("sa_intv", C.c_int32), ("n_sa", C.c_uint64), ("sa", C.c_void_p)]


def make_cbwt(primary, L2, seq_len, bwt_words, sa_intv, sa, keep):
    """bmh_bwt_t over numpy arrays (kept alive through `keep`)."""
    bw = np.ascontiguousarray(bwt_words, dtype=np.uint32)
    sa = np.ascontiguousarray(sa, dtype=np.uint64)
    keep += [bw, sa]
    b = CBwt()
    b.primary, b.seq_len, b.bwt_size, b.sa_intv, b.n_sa = int(primary), int(seq_len), len(bw), int(sa_intv), len(sa)
    for i in range(5):
        b.L2[i] = int(L2[i])
    b.bwt, b.sa = bw.ctypes.data, sa.ctypes.data
    return b


def smem_opt(opt):
    """A bmh_smem_opt_t record from one that may predate a field (the fixtures hold the first four): missing fields are 0."""
    opt = np.asarray(opt)
    out = np.zeros((), dtype=SMEM_OPT)
    for k in opt.dtype.names:
        out[k] = opt[k]
    return out


def orc_smem_calls(cb, opt, read):
    """OUR restatement of the bwt_smem1 call sequence of one read -> (SMEM_CALL[], SMEM_INTV[])."""
    lib = load_oracle()
    read = np.ascontiguousarray(read, dtype=np.uint8)
    L = len(read)
    calls = np.zeros(2 * L + 4, dtype=SMEM_CALL)
    pool = np.zeros(4 * (L + 2) * 4 + 64, dtype=SMEM_INTV)
    used = C.c_int(0)
    oo = smem_opt(opt)
    lib.orc_smem_calls.restype = C.c_int
    n = lib.orc_smem_calls(C.byref(cb), oo.ctypes.data_as(C.c_void_p), C.c_int(L), read.ctypes.data_as(C.c_void_p),
                           calls.ctypes.data_as(C.c_void_p), C.c_int(len(calls)), pool.ctypes.data_as(C.c_void_p),
                           C.c_int(len(pool)), C.byref(used))
    assert n >= 0
    return calls[:n].copy(), pool[:used.value].copy()


def orc_sa(cb, ks):
    lib = load_oracle()
    lib.orc_bwt_sa.restype = C.c_uint64
    return np.array([lib.orc_bwt_sa(C.byref(cb), C.c_uint64(int(k))) for k in ks], dtype=np.uint64)


def golden_fmindex():
    """(bmh_bwt_t over the fixture's index, raw arrays, opt, reads, per-read (calls, intervals), sa_k, sa_pos)."""
    g = load_golden("fmindex_golden.npz")
    keep = []
    cb = make_cbwt(int(g["primary"]), g["L2"], int(g["seq_len"]), g["bwt"], int(g["sa_intv"]), g["sa"], keep)
    raw = (int(g["primary"]), [int(x) for x in g["L2"]], int(g["seq_len"]), g["bwt"], int(g["sa_intv"]), g["sa"])
    reads, per, o, c0, i0 = [], [], 0, 0, 0
    for L, cn, inn in zip(g["read_len"], g["call_n"], g["intv_n"]):
        reads.append(g["reads"][o:o + L].copy())
        per.append((g["calls"][c0:c0 + cn].copy(), g["intv"][i0:i0 + inn].copy()))
        o, c0, i0 = o + L, c0 + cn, i0 + inn
    return cb, keep, raw, smem_opt(g["opt"]), reads, per, g["sa_k"], g["sa_pos"]
