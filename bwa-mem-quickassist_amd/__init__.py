"""bwa-mem-quickassist_amd -- ctypes binding of libbwamem_hip.so (include/bwamem_hip.h).

The product is the C-ABI shared library built from csrc/ (HIP kernels for gfx950) and
host/ (the batched extension driver, plain C).  This module only loads it and gives the
tests and bench.py a thin, numpy/torch-friendly view of the same entry points; it holds
no algorithmic code and NO fallback: if the library is missing or no GPU is usable,
every call raises.

The directory name is fixed by the project layout and is not a valid Python identifier;
load it with `__graft_entry__.load_package()` (importlib, module name
`bwa_mem_quickassist_amd`).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbwamem_hip.so")
DROPIN_PATH = os.path.join(_HERE, "libbwamem_hip_dropin.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "bwamem_hip.h")

BMH_OK, BMH_E_NODEVICE, BMH_E_HIP, BMH_E_ARG, BMH_E_RANGE, BMH_E_NOMEM, BMH_E_CIGAR_CAP = 0, -1, -2, -3, -4, -5, -6
BMH_F_QREV, BMH_F_TREV, BMH_F_TPAC, BMH_F_QCOMP = 1, 2, 4, 8

# record layouts == include/bwamem_hip.h
EXT_TASK = np.dtype([("q_off", "<u8"), ("t_off", "<u8"), ("qlen", "<u2"), ("tlen", "<u2"),
                     ("h0", "<i4"), ("w", "<i2"), ("end_bonus", "<i2"), ("flags", "<u2"),
                     ("rsv", "<u2")])
EXT_RES = np.dtype([("score", "<i4"), ("qle", "<i4"), ("tle", "<i4"), ("gtle", "<i4"),
                    ("gscore", "<i4"), ("max_off", "<i4")])
SEED_TASK = np.dtype([("q_off", "<u8"), ("t_off", "<u8"), ("l_query", "<i4"), ("qbeg", "<i4"), ("len", "<i4"),
                      ("rbeg", "<i4"), ("wlen", "<i4"), ("flags", "<u2"), ("rsv", "<u2")])
SEED_RES = np.dtype([("qb", "<i4"), ("qe", "<i4"), ("rb", "<i4"), ("re", "<i4"), ("score", "<i4"), ("truesc", "<i4"),
                     ("w", "<i4"), ("n_ext", "<i4")])
GLB_TASK = np.dtype([("q_off", "<u8"), ("t_off", "<u8"), ("qlen", "<u2"), ("tlen", "<u2"),
                     ("w", "<i4"), ("cigar_off", "<u4"), ("cigar_cap", "<u4")])
GLB_RES = np.dtype([("score", "<i4"), ("n_cigar", "<i4")])
SW_TASK = np.dtype([("q_off", "<u8"), ("t_off", "<u8"), ("tlen", "<u4"), ("qlen", "<u2"), ("flags", "<u2"),
                    ("xtra", "<u4"), ("rsv", "<u4")])
SW_RES = np.dtype([("score", "<i4"), ("te", "<i4"), ("qe", "<i4"), ("score2", "<i4"), ("te2", "<i4"),
                   ("tb", "<i4"), ("qb", "<i4"), ("rsv", "<i4")])
KSW_XBYTE, KSW_XSTOP, KSW_XSUBO, KSW_XSTART = 0x10000, 0x20000, 0x40000, 0x80000  # reference ksw.h:6-9
SMEM_INTV = np.dtype([("x0", "<u8"), ("x1", "<u8"), ("x2", "<u8"), ("info", "<u8")])
SMEM_CALL = np.dtype([("x", "<i4"), ("min_intv", "<i4"), ("ret", "<i4"), ("n", "<i4"), ("first", "<u4"), ("rsv", "<u4")])
SMEM_OPT = np.dtype([("min_seed_len", "<i4"), ("split_len", "<i4"), ("split_width", "<i4"), ("start_width", "<i4"), ("min_emit_len", "<i4")])


class _Bwt(C.Structure):  # bmh_bwt_t
    _fields_ = [("primary", C.c_uint64), ("L2", C.c_uint64 * 5), ("seq_len", C.c_uint64), ("bwt_size", C.c_uint64),
                ("bwt", C.c_void_p), ("sa_intv", C.c_int32), ("n_sa", C.c_uint64), ("sa", C.c_void_p)]


PESTAT = np.dtype([("low", "<i4"), ("high", "<i4"), ("failed", "<i4"), ("pad", "<i4"), ("avg", "<f8"), ("std", "<f8")])
MATESW_OPT = np.dtype([("pen_unpaired", "<i4"), ("max_matesw", "<i4"), ("min_seed_len", "<i4"), ("rsv", "<i4")])
PARAMS = np.dtype([("o_del", "<i4"), ("e_del", "<i4"), ("o_ins", "<i4"), ("e_ins", "<i4"),
                   ("zdrop", "<i4"), ("a", "<i4"), ("w", "<i4"), ("pen_clip5", "<i4"),
                   ("pen_clip3", "<i4"), ("mat", "i1", (25,)), ("pad", "i1", (3,))])
SEED = np.dtype([("rbeg", "<i8"), ("qbeg", "<i4"), ("len", "<i4")])
CIGAR_REQ = np.dtype([("read", "<i4"), ("qb", "<i4"), ("qe", "<i4"), ("pad", "<i4"), ("rb", "<i8"), ("re", "<i8"),
                      ("truesc", "<i4"), ("reg_w", "<i4")])
CIGAR_RES = np.dtype([("score", "<i4"), ("n_cigar", "<i4"), ("NM", "<i4"), ("tries", "<i4"), ("cigar_off", "<u4"),
                      ("md_off", "<u4"), ("md_len", "<u4"), ("rsv", "<u4")])
BMH_REGION_CIGAR_CUT, BMH_REGION_MD_CUT = 1, 2
REGION_REQ = np.dtype([("q_src", "<u8"), ("rb", "<i8"), ("o_off", "<u8"), ("ql", "<i4"), ("tl", "<i4"), ("truesc", "<i4"),
                       ("task", "<i4", (3,))])
REGION_RES = np.dtype([("score", "<i4"), ("n_cigar", "<i4"), ("tries", "<i4"), ("NM", "<i4"), ("md_len", "<i4"), ("flags", "<u4")])
ALNREG = np.dtype([("rb", "<i8"), ("re", "<i8"), ("qb", "<i4"), ("qe", "<i4"), ("score", "<i4"),
                   ("truesc", "<i4"), ("sub", "<i4"), ("csub", "<i4"), ("sub_n", "<i4"),
                   ("w", "<i4"), ("seedcov", "<i4"), ("secondary", "<i4"), ("hash", "<u8")])


class BmhError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        super().__init__(f"libbwamem_hip error {code}: {detail}")


class _Chain(C.Structure):
    _fields_ = [("n", C.c_int32), ("m", C.c_int32), ("pos", C.c_int64), ("seeds", C.c_void_p)]


class _ChainV(C.Structure):
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("a", C.POINTER(_Chain))]


class _AlnregV(C.Structure):
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("a", C.c_void_p)]


class _Read(C.Structure):
    _fields_ = [("l_seq", C.c_int32), ("seq", C.c_void_p)]


class _DriverStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("rounds", "ext_tasks", "seeds_extended", "seeds_skipped", "pool_bytes", "seeds_speculated", "short_sw")]


_lib = None


def lib():
    """The loaded C-ABI library.  Raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BmhError(BMH_E_NODEVICE, f"{LIB_PATH} not built; run __graft_entry__.build()")
        L = C.CDLL(LIB_PATH)
        L.bmh_strerror.restype = C.c_char_p
        L.bmh_last_error.restype = C.c_char_p
        L.bmh_last_error.argtypes = [C.c_void_p]
        for name in ("bmh_ctx_destroy", "bmh_ctx_sync"):
            getattr(L, name).argtypes = [C.c_void_p]
        L.bmh_ctx_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        L.bmh_ctx_set_params.argtypes = [C.c_void_p, C.c_void_p]
        L.bmh_ctx_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.bmh_ctx_set_qcap.argtypes = [C.c_void_p, C.c_int]
        L.bmh_ctx_reserve_staging.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
        L.bmh_ctx_set_pac.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.bmh_set_kernel_timing.argtypes = [C.c_void_p, C.c_int]
        L.bmh_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.bmh_last_extend_bin_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.bmh_last_global_bin_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.bmh_last_seedext_round_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.bmh_extend_bin_ms_sum.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.c_int]
        L.bmh_upload_pool.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.bmh_extend_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int64, C.c_void_p]
        L.bmh_extend_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.bmh_seedext_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int64, C.c_void_p]
        L.bmh_seedext_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.bmh_seedext_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.bmh_seedext_wait.argtypes = [C.c_void_p, C.c_void_p]
        L.bmh_seedext_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.bmh_ctx_set_bwt.argtypes = [C.c_void_p, C.c_void_p]
        L.bmh_smem_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                     C.c_void_p, C.c_size_t]
        L.bmh_sa_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.bmh_matesw_batch.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.bmh_sw_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int64, C.c_void_p]
        L.bmh_sw_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.bmh_extend_batch_sharded.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_size_t, C.c_void_p,
                                               C.c_int64, C.c_void_p]
        L.bmh_global_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int64, C.c_void_p,
                                       C.c_void_p, C.c_size_t]
        L.bmh_global_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                              C.c_void_p]
        L.bmh_chain2aln_batch.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]
        L.bmh_driver_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.bmh_region_cigar_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                             C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.bmh_region_cigar_batch.restype = C.c_int
        L.bmh_reg2cigar_batch.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        _lib = L
    return _lib


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]
_libc.malloc.restype = C.c_void_p
_libc.malloc.argtypes = [C.c_size_t]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Context:
    """One GPU + one stream (bmh_ctx_t).  Mirrors the C-ABI one to one."""

    def __init__(self, device=0, params=None):
        self._h = C.c_void_p()
        rc = lib().bmh_ctx_create(C.byref(self._h), int(device))
        if rc:
            raise BmhError(rc, lib().bmh_strerror(rc).decode())
        if params is not None:
            self.set_params(params)

    def _check(self, rc):
        if rc:
            raise BmhError(rc, f"{lib().bmh_strerror(rc).decode()}: {lib().bmh_last_error(self._h).decode()}")

    def close(self):
        if self._h:
            lib().bmh_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, p):
        p = np.ascontiguousarray(np.asarray(p, dtype=PARAMS).reshape(()))
        self._check(lib().bmh_ctx_set_params(self._h, _ptr(p)))

    def set_stream(self, hip_stream_ptr):
        self._check(lib().bmh_ctx_set_stream(self._h, C.c_void_p(hip_stream_ptr)))

    def set_pac(self, pac, l_pac):
        """Make the 2-bit reference resident on the device (BMH_F_TPAC tasks; drivers skip the host bns_get_seq).
        The array is kept alive and must be passed unchanged to the drivers (they recognise it by address)."""
        pac = np.ascontiguousarray(pac, dtype=np.uint8)
        self._pac = pac
        self._check(lib().bmh_ctx_set_pac(self._h, _ptr(pac), int(l_pac)))
        return pac

    def set_qcap(self, q):
        self._check(lib().bmh_ctx_set_qcap(self._h, int(q)))

    def reserve_staging(self, upload_bytes, download_bytes):
        """Pinned staging buffers of the host-buffer entry points, allocated ahead of the first batch."""
        self._check(lib().bmh_ctx_reserve_staging(self._h, int(upload_bytes), int(download_bytes)))

    def sync(self):
        self._check(lib().bmh_ctx_sync(self._h))

    def set_kernel_timing(self, on=True):
        self._check(lib().bmh_set_kernel_timing(self._h, 1 if on else 0))

    def last_kernel_ms(self):
        ms = C.c_float(-1)
        self._check(lib().bmh_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def last_extend_bin_ms(self):
        ms = (C.c_float * 6)()
        self._check(lib().bmh_last_extend_bin_ms(self._h, ms))
        return [float(x) for x in ms]

    def extend_bin_ms_sum(self, reset=True):
        """per-bin kernel time summed over the dispatcher launches since the last reset (timing mode) -> (ms[6], launches)"""
        ms = (C.c_double * 6)()
        n = C.c_longlong(0)
        self._check(lib().bmh_extend_bin_ms_sum(self._h, ms, C.byref(n), 1 if reset else 0))
        return [float(x) for x in ms], int(n.value)

    def last_seedext_round_ms(self):
        ms = (C.c_float * 4)()
        self._check(lib().bmh_last_seedext_round_ms(self._h, ms))
        return [float(x) for x in ms]

    def last_global_bin_ms(self):
        ms = (C.c_float * 3)()
        self._check(lib().bmh_last_global_bin_ms(self._h, ms))
        return [float(x) for x in ms]

    # ---- L2, host buffers
    def extend_batch(self, pool, tasks):
        """N x ksw_extend2 (reference ksw.c:379).  numpy in, numpy out."""
        pool = np.ascontiguousarray(pool, dtype=np.uint8)
        tasks = np.ascontiguousarray(tasks, dtype=EXT_TASK)
        res = np.zeros(len(tasks), dtype=EXT_RES)
        self._check(lib().bmh_extend_batch(self._h, _ptr(pool), pool.nbytes, _ptr(tasks), len(tasks), _ptr(res)))
        return res

    def seedext_batch(self, pool, tasks):
        """N x (left extension, clip decision, right extension from the left score): the fused per-seed record
        (reference bwamem.c:810-866; ext_param_t / ext_res_t :553-577).  numpy in, numpy out."""
        pool = np.ascontiguousarray(pool, dtype=np.uint8)
        tasks = np.ascontiguousarray(tasks, dtype=SEED_TASK)
        res = np.zeros(len(tasks), dtype=SEED_RES)
        self._check(lib().bmh_seedext_batch(self._h, _ptr(pool), pool.nbytes, _ptr(tasks), len(tasks), _ptr(res)))
        return res

    def seedext_batch_device(self, d_pool, d_tasks, n, d_res):
        self._check(lib().bmh_seedext_batch_device(self._h, C.c_void_p(d_pool), C.c_void_p(d_tasks), int(n), C.c_void_p(d_res)))

    def seedext_stats(self):
        st = (C.c_int64 * 5)()
        self._check(lib().bmh_seedext_stats(self._h, st))
        return dict(zip(("seeds", "left_tasks", "left_retries", "right_tasks", "right_retries"), [int(x) for x in st]))

    def global_batch(self, pool, tasks, cigar_words):
        """N x ksw_global2 (reference ksw.c:501).  Returns (results, cigar_pool)."""
        pool = np.ascontiguousarray(pool, dtype=np.uint8)
        tasks = np.ascontiguousarray(tasks, dtype=GLB_TASK)
        res = np.zeros(len(tasks), dtype=GLB_RES)
        cig = np.zeros(max(int(cigar_words), 1), dtype=np.uint32)
        self._check(lib().bmh_global_batch(self._h, _ptr(pool), pool.nbytes, _ptr(tasks), len(tasks), _ptr(res),
                                           _ptr(cig), int(cigar_words)))
        return res, cig

    def region_cigar_batch(self, readpool, opool_bytes, reqs, tasks, task_cigar_words, cig_cap=24, md_cap=64):
        """One record per region: bwa_gen_cigar2's byte work around ksw_global2 on the device (reference bwa.c:89-172, bwamem.c:1194-1201);
        needs set_pac().  Returns (results, cigar words [n, cig_cap], MD bytes [n, md_cap])."""
        readpool = np.ascontiguousarray(readpool, dtype=np.uint8)
        reqs = np.ascontiguousarray(reqs, dtype=REGION_REQ)
        tasks = np.ascontiguousarray(tasks, dtype=GLB_TASK)
        res = np.zeros(len(reqs), dtype=REGION_RES)
        cig = np.zeros((max(len(reqs), 1), cig_cap), dtype=np.uint32)
        md = np.zeros((max(len(reqs), 1), md_cap), dtype=np.uint8)
        self._check(lib().bmh_region_cigar_batch(self._h, _ptr(readpool), readpool.nbytes, int(opool_bytes), _ptr(reqs), len(reqs),
                                                 _ptr(tasks), len(tasks), int(task_cigar_words), int(cig_cap), int(md_cap), _ptr(res),
                                                 _ptr(cig), _ptr(md)))
        return res, cig, md

    # ---- L2, device-resident (raw device pointers, e.g. torch tensors' data_ptr())
    def sw_batch(self, pool, tasks):
        """N x ksw_align2 (reference ksw.c:341; mate rescue / short chains).  numpy in, numpy out (kswr_t fields)."""
        pool = np.ascontiguousarray(pool, dtype=np.uint8)
        tasks = np.ascontiguousarray(tasks, dtype=SW_TASK)
        res = np.zeros(len(tasks), dtype=SW_RES)
        self._check(lib().bmh_sw_batch(self._h, _ptr(pool), pool.nbytes, _ptr(tasks), len(tasks), _ptr(res)))
        return res

    def sw_batch_device(self, d_pool, d_tasks, n, d_res):
        self._check(lib().bmh_sw_batch_device(self._h, C.c_void_p(d_pool), C.c_void_p(d_tasks), int(n), C.c_void_p(d_res)))

    def set_bwt(self, primary, L2, seq_len, bwt_words, sa_intv, sa):
        """Make the FM-index resident (the arrays of the reference's bwt_t, bwt.h:45-57)."""
        bw = np.ascontiguousarray(bwt_words, dtype=np.uint32)
        sa = np.ascontiguousarray(sa, dtype=np.uint64)
        self._bwt_keep = (bw, sa)
        b = _Bwt()
        b.primary, b.seq_len, b.bwt_size, b.sa_intv, b.n_sa = int(primary), int(seq_len), len(bw), int(sa_intv), len(sa)
        for i in range(5):
            b.L2[i] = int(L2[i])
        b.bwt, b.sa = bw.ctypes.data, sa.ctypes.data
        self._check(lib().bmh_ctx_set_bwt(self._h, C.byref(b)))

    def smem_batch(self, opt, reads):
        """The bwt_smem1 calls of smem_next2's iteration for every read (reference bwt.c:288, bwamem.c:118).
        Returns per read (SMEM_CALL[], SMEM_INTV[])."""
        n = len(reads)
        o_in, opt = np.asarray(opt), np.zeros((), dtype=SMEM_OPT)
        for k in o_in.dtype.names:  # (records written before a field existed leave it 0)
            opt[k] = o_in[k]
        keep = []
        c_reads = (_Read * max(n, 1))()
        tot = 0
        for k, r in enumerate(reads):
            r = np.ascontiguousarray(r, dtype=np.uint8)
            keep.append(r)
            c_reads[k].l_seq, c_reads[k].seq = len(r), r.ctypes.data
            tot += len(r)
        call_cap, intv_cap = tot // 4 + 64 * n + 64, 2 * tot + 1024
        call_off = np.zeros(n + 1, dtype=np.uint32)
        intv_off = np.zeros(n + 1, dtype=np.uint64)
        while True:
            calls = np.zeros(call_cap, dtype=SMEM_CALL)
            intv = np.zeros(intv_cap, dtype=SMEM_INTV)
            rc = lib().bmh_smem_batch(self._h, _ptr(opt), n, C.cast(c_reads, C.c_void_p), _ptr(call_off), _ptr(calls),
                                      C.c_size_t(call_cap), _ptr(intv_off), _ptr(intv), C.c_size_t(intv_cap))
            if rc != BMH_E_CIGAR_CAP:
                break
            call_cap, intv_cap = 2 * call_cap, 4 * intv_cap  # totals are data dependent: grow and ask again
        self._check(rc)
        out = []
        for r in range(n):
            out.append((calls[call_off[r]:call_off[r + 1]].copy(), intv[int(intv_off[r]):int(intv_off[r + 1])].copy()))
        return out

    def seed_batch(self, opt, max_occ, reads):
        """bmh_seed_batch: smem_batch plus, per interval, where its suffix-array positions start (sa_off, UINT64_MAX = not
        looked up) and the positions.  Returns (per-read [(calls, intervals)], per-read sa_off arrays, sa_pos)."""
        n = len(reads)
        o_in, opt = np.asarray(opt), np.zeros((), dtype=SMEM_OPT)
        for k in o_in.dtype.names:
            opt[k] = o_in[k]
        keep = []
        c_reads = (_Read * max(n, 1))()
        tot = 0
        for k, r in enumerate(reads):
            r = np.ascontiguousarray(r, dtype=np.uint8)
            keep.append(r)
            c_reads[k].l_seq, c_reads[k].seq = len(r), r.ctypes.data
            tot += len(r)
        call_cap, intv_cap, sa_cap = tot // 4 + 64 * n + 64, 2 * tot + 1024, 4 * tot + 4096
        call_off = np.zeros(n + 1, dtype=np.uint32)
        intv_off = np.zeros(n + 1, dtype=np.uint64)
        n_pos = C.c_uint64(0)
        while True:
            calls = np.zeros(call_cap, dtype=SMEM_CALL)
            intv = np.zeros(intv_cap, dtype=SMEM_INTV)
            sa_off = np.zeros(intv_cap, dtype=np.uint64)
            sa_pos = np.zeros(sa_cap, dtype=np.uint64)
            rc = lib().bmh_seed_batch(self._h, _ptr(opt), C.c_int(int(max_occ)), n, C.cast(c_reads, C.c_void_p), _ptr(call_off), _ptr(calls),
                                      C.c_size_t(call_cap), _ptr(intv_off), _ptr(intv), C.c_size_t(intv_cap), _ptr(sa_off), _ptr(sa_pos),
                                      C.c_size_t(sa_cap), C.byref(n_pos))
            if rc != BMH_E_CIGAR_CAP:
                break
            call_cap, intv_cap, sa_cap = 2 * call_cap, 4 * intv_cap, 4 * sa_cap
        self._check(rc)
        out, offs = [], []
        for r in range(n):
            lo, hi = int(intv_off[r]), int(intv_off[r + 1])
            out.append((calls[call_off[r]:call_off[r + 1]].copy(), intv[lo:hi].copy()))
            offs.append(sa_off[lo:hi].copy())
        return out, offs, sa_pos[:n_pos.value].copy()

    def sa_batch(self, ks):
        """N x bwt_sa (reference bwt.c:85)."""
        ks = np.ascontiguousarray(ks, dtype=np.uint64)
        pos = np.zeros(len(ks), dtype=np.uint64)
        self._check(lib().bmh_sa_batch(self._h, _ptr(ks), len(ks), _ptr(pos)))
        return pos

    def matesw_batch(self, l_pac, pac, reads, regs, pes, opt, dedup):
        """Batched mate rescue (reference bwamem_pair.c:251-263 over mem_matesw :109-175) for len(reads)//2 pairs.
        reads: flat list of uint8 code arrays (2 per pair); regs: flat list of ALNREG arrays; pes: PESTAT[4];
        opt: MATESW_OPT record; dedup: C function pointer with the bmh_dedup_fn shape.
        Returns (regs after rescue, n per pair)."""
        n_pairs = len(reads) // 2
        pac = np.ascontiguousarray(pac, dtype=np.uint8)
        pes = np.ascontiguousarray(pes, dtype=PESTAT)
        opt = np.ascontiguousarray(opt, dtype=MATESW_OPT)
        keep = []
        c_reads = (_Read * len(reads))()
        for k, r in enumerate(reads):
            r = np.ascontiguousarray(r, dtype=np.uint8)
            keep.append(r)
            c_reads[k].l_seq, c_reads[k].seq = len(r), r.ctypes.data
        c_regs = (_AlnregV * len(regs))()
        for k, r in enumerate(regs):
            r = np.ascontiguousarray(r, dtype=ALNREG)
            c_regs[k].n = c_regs[k].m = len(r)
            if len(r):
                c_regs[k].a = _libc.malloc(len(r) * ALNREG.itemsize)
                C.memmove(c_regs[k].a, r.ctypes.data, len(r) * ALNREG.itemsize)
        n_sw = np.zeros(max(n_pairs, 1), dtype=np.int32)
        rc = lib().bmh_matesw_batch(self._h, C.c_int64(l_pac), _ptr(pac), n_pairs, C.cast(c_reads, C.c_void_p),
                                    C.cast(c_regs, C.c_void_p), _ptr(pes), _ptr(opt), dedup, None, _ptr(n_sw))
        out = []
        for k in range(len(regs)):
            a = np.zeros(c_regs[k].n, dtype=ALNREG)
            if c_regs[k].n:
                C.memmove(a.ctypes.data, c_regs[k].a, c_regs[k].n * ALNREG.itemsize)
            if c_regs[k].a:
                _libc.free(c_regs[k].a)
            out.append(a)
        self._check(rc)
        return out, n_sw[:n_pairs].tolist()

    def extend_batch_device(self, d_pool, d_tasks, n, d_res, d_order=0):
        self._check(lib().bmh_extend_batch_device(self._h, C.c_void_p(d_pool), C.c_void_p(d_tasks), int(n),
                                                  C.c_void_p(d_res), C.c_void_p(d_order) if d_order else None))

    def global_batch_device(self, d_pool, d_tasks, n, d_res, d_cigar, d_order=0):
        self._check(lib().bmh_global_batch_device(self._h, C.c_void_p(d_pool), C.c_void_p(d_tasks), int(n),
                                                  C.c_void_p(d_res), C.c_void_p(d_cigar),
                                                  C.c_void_p(d_order) if d_order else None))

    # ---- L3 driver
    def chain2aln_batch(self, l_pac, pac, reads, chains):
        """reads: list of uint8 code arrays; chains: per read a list of SEED arrays.
        Returns per read an ALNREG array (what mem_chain2aln appends, reference bwamem.c:730-878)."""
        n = len(reads)
        pac = np.ascontiguousarray(pac, dtype=np.uint8)
        keep = []
        c_reads = (_Read * n)()
        c_chv = (_ChainV * n)()
        c_regs = (_AlnregV * n)()
        for r in range(n):
            seq = np.ascontiguousarray(reads[r], dtype=np.uint8)
            keep.append(seq)
            c_reads[r].l_seq, c_reads[r].seq = len(seq), seq.ctypes.data
            arr = (_Chain * max(len(chains[r]), 1))()
            for ci, seeds in enumerate(chains[r]):
                sd = np.ascontiguousarray(seeds, dtype=SEED)
                keep.append(sd)
                arr[ci].n = arr[ci].m = len(sd)
                arr[ci].pos = int(sd["rbeg"][0]) if len(sd) else 0
                arr[ci].seeds = sd.ctypes.data
            keep.append(arr)
            c_chv[r].n = c_chv[r].m = len(chains[r])
            c_chv[r].a = C.cast(arr, C.POINTER(_Chain))
        rc = lib().bmh_chain2aln_batch(self._h, int(l_pac), _ptr(pac), n, C.cast(c_reads, C.c_void_p),
                                       C.cast(c_chv, C.c_void_p), None, None, C.cast(c_regs, C.c_void_p))
        out = []
        for r in range(n):
            k = c_regs[r].n
            a = np.zeros(k, dtype=ALNREG)
            if k:
                C.memmove(a.ctypes.data, c_regs[r].a, k * ALNREG.itemsize)
            if c_regs[r].a:
                _libc.free(c_regs[r].a)
            out.append(a)
        self._check(rc)
        return out

    def reg2cigar_batch(self, l_pac, pac, reads, reqs):
        """Batched mem_reg2aln band/retry loop over bwa_gen_cigar2 (reference bwamem.c:1187-1201, bwa.c:89-172).
        reads: list of uint8 code arrays; reqs: CIGAR_REQ array.  Returns (results, cigar_pool, md_bytes)."""
        pac = np.ascontiguousarray(pac, dtype=np.uint8)
        reqs = np.ascontiguousarray(reqs, dtype=CIGAR_REQ)
        keep = []
        c_reads = (_Read * max(len(reads), 1))()
        for r, seq in enumerate(reads):
            seq = np.ascontiguousarray(seq, dtype=np.uint8)
            keep.append(seq)
            c_reads[r].l_seq, c_reads[r].seq = len(seq), seq.ctypes.data
        span = (reqs["qe"] - reqs["qb"]).astype(np.int64) + (reqs["re"] - reqs["rb"]).astype(np.int64)
        span = np.maximum(span, 0)
        cw, mb = int(span.sum() + 2 * len(reqs) + 8), int(3 * span.sum() + 16 * len(reqs) + 16)
        res = np.zeros(len(reqs), dtype=CIGAR_RES)
        cig = np.zeros(cw, dtype=np.uint32)
        md = np.zeros(mb, dtype=np.uint8)
        self._check(lib().bmh_reg2cigar_batch(self._h, int(l_pac), _ptr(pac), C.cast(c_reads, C.c_void_p), len(reqs),
                                              _ptr(reqs), _ptr(res), _ptr(cig), cw, _ptr(md), mb))
        return res, cig, md

    def driver_stats(self):
        st = _DriverStats()
        self._check(lib().bmh_driver_stats(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in _DriverStats._fields_}


def extend_batch_sharded(ctxs, pool, tasks):
    """Static contiguous shard of one host batch over several contexts (one per GPU)."""
    pool = np.ascontiguousarray(pool, dtype=np.uint8)
    tasks = np.ascontiguousarray(tasks, dtype=EXT_TASK)
    res = np.zeros(len(tasks), dtype=EXT_RES)
    arr = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    rc = lib().bmh_extend_batch_sharded(arr, len(ctxs), _ptr(pool), pool.nbytes, _ptr(tasks), len(tasks), _ptr(res))
    if rc:
        raise BmhError(rc, lib().bmh_strerror(rc).decode())
    return res


def _sharded(fn, ctxs, pool, tasks, tdtype, rdtype, *more):
    pool = np.ascontiguousarray(pool, dtype=np.uint8)
    tasks = np.ascontiguousarray(tasks, dtype=tdtype)
    res = np.zeros(len(tasks), dtype=rdtype)
    arr = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    rc = getattr(lib(), fn)(arr, len(ctxs), _ptr(pool), C.c_size_t(pool.nbytes), _ptr(tasks), C.c_int64(len(tasks)), _ptr(res), *more)
    if rc:
        raise BmhError(rc, lib().bmh_strerror(rc).decode())
    return res


def seedext_batch_sharded(ctxs, pool, tasks):
    """Fused per-seed records, one contiguous slice per context (bmh_seedext_batch_sharded)."""
    return _sharded("bmh_seedext_batch_sharded", ctxs, pool, tasks, SEED_TASK, SEED_RES)


def sw_batch_sharded(ctxs, pool, tasks):
    """ksw_align2 tasks, one contiguous slice per context (bmh_sw_batch_sharded)."""
    return _sharded("bmh_sw_batch_sharded", ctxs, pool, tasks, SW_TASK, SW_RES)


def global_batch_sharded(ctxs, pool, tasks, cigar_words):
    """ksw_global2 + traceback, one contiguous slice per context (bmh_global_batch_sharded) -> (results, CIGAR pool)."""
    cig = np.zeros(max(int(cigar_words), 1), dtype=np.uint32)
    res = _sharded("bmh_global_batch_sharded", ctxs, pool, tasks, GLB_TASK, GLB_RES, _ptr(cig), C.c_size_t(int(cigar_words)))
    return res, cig


def declared_symbols():
    """Every function name declared in include/bwamem_hip.h (for the symbol-export test)."""
    import re
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(bmh_[a-z0-9_]+)\s*\(", txt)))
