"""ctypes access to the REFERENCE compiled by oracle/Makefile (oracle/_ref/libbwa_ref.so):
index loading, seeding+chaining and the reference's own mem_chain2aln.  Checker-side only;
used by tools/make_golden.py and by the `ref`-gated tests.  Struct mirrors cite the reference."""
import ctypes as C
import os
import subprocess

import numpy as np

import kswlib

REF_BWA = os.path.join(kswlib.REF_DIR, "bwa")
REF_LIB = os.path.join(kswlib.REF_DIR, "libbwa_ref.so")


def have_ref_bwa():
    return os.path.exists(REF_BWA) and os.path.exists(REF_LIB)


class MemOpt(C.Structure):  # fork's mem_opt_t, bwamem.h:21-48
    _fields_ = [(n, C.c_int) for n in ("a", "b", "o_del", "e_del", "o_ins", "e_ins", "pen_unpaired", "pen_clip5",
                                        "pen_clip3", "w", "zdrop", "T", "flag", "min_seed_len")] + \
               [("split_factor", C.c_float)] + \
               [(n, C.c_int) for n in ("split_width", "max_occ", "max_chain_gap", "n_threads", "batch_size",
                                        "chunk_size")] + \
               [(n, C.c_float) for n in ("mask_level", "chain_drop_ratio", "mask_level_redun", "mapQ_coef_len")] + \
               [(n, C.c_int) for n in ("mapQ_coef_fac", "max_ins", "max_matesw")] + [("mat", C.c_int8 * 25)]


class Seed(C.Structure):  # mem_seed_t, bwamem.c:168-171
    _fields_ = [("rbeg", C.c_int64), ("qbeg", C.c_int32), ("len", C.c_int32)]


class Chain(C.Structure):  # mem_chain_t, bwamem.c:173-177
    _fields_ = [("n", C.c_int), ("m", C.c_int), ("pos", C.c_int64), ("seeds", C.POINTER(Seed))]


class ChainV(C.Structure):  # mem_chain_v, bwamem.c:179
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("a", C.POINTER(Chain))]


class AlnregV(C.Structure):  # mem_alnreg_v, bwamem.h:64
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("a", C.c_void_p)]


class BntSeqHead(C.Structure):  # first field of bntseq_t, bntseq.h:53
    _fields_ = [("l_pac", C.c_int64)]


class BwaIdx(C.Structure):  # bwaidx_t, bwa.h:13-17
    _fields_ = [("bwt", C.c_void_p), ("bns", C.POINTER(BntSeqHead)), ("pac", C.POINTER(C.c_uint8))]


_lib = None
_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(REF_LIB)
        L.mem_opt_init.restype = C.POINTER(MemOpt)
        L.bwa_idx_load.restype = C.POINTER(BwaIdx)
        L.bwa_idx_load.argtypes = [C.c_char_p, C.c_int]
        L.mem_chain.restype = ChainV
        L.mem_chain.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        L.mem_chain_flt.restype = C.c_int
        L.mem_chain_flt.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.mem_chain2aln.restype = None
        L.mem_chain2aln.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.bwa_fill_scmat.argtypes = [C.c_int, C.c_int, C.c_void_p]
        C.c_int.in_dll(L, "bwa_verbose").value = 1
        _lib = L
    return _lib


def write_fasta(path, name, codes):
    """one contig, 80 bases per line (vectorised: a 1 Gb genome is written in seconds)"""
    codes = np.asarray(codes, dtype=np.uint8)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
    with open(path, "wb") as f:
        f.write(f">{name}\n".encode())
        full = len(letters) // 80 * 80
        if full:
            body = np.empty((full // 80, 81), dtype=np.uint8)
            body[:, :80] = letters[:full].reshape(-1, 80)
            body[:, 80] = 10
            body.tofile(f)
        if len(letters) > full:
            f.write(letters[full:].tobytes() + b"\n")


def write_fastq(path, reads, prefix="r"):
    with open(path, "w") as f:
        for i, r in enumerate(reads):
            f.write(f"@{prefix}{i}\n" + "".join("ACGTN"[c] for c in r) + "\n+\n" + "I" * len(r) + "\n")


def build_index(fasta, log=None):
    """`bwa index` of the compiled reference; log: file that receives its progress lines (a long build must show signs of life)"""
    if log:
        with open(log, "w") as f:
            subprocess.run([REF_BWA, "index", fasta], check=True, stdout=f, stderr=subprocess.STDOUT)
    else:
        subprocess.run([REF_BWA, "index", fasta], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def opt_from_params(p):
    """mem_opt_t with the hot-path fields taken from a kswlib.PARAMS record."""
    L = lib()
    o = L.mem_opt_init()
    for k in ("a", "o_del", "e_del", "o_ins", "e_ins", "w", "zdrop", "pen_clip5", "pen_clip3"):
        setattr(o.contents, k, int(p[k]))
    for i in range(25):
        o.contents.mat[i] = int(p["mat"][i])
    return o


def chains_and_regs(idx, opt, reads, run_chain2aln=True):
    """Per read: the reference's chains after mem_chain + mem_chain_flt (bwamem.c:1131-1132) and, if asked,
    the regions the reference's OWN mem_chain2aln appends for them (bwamem.c:1141, shared vector)."""
    L = lib()
    l_pac = idx.contents.bns.contents.l_pac
    all_chains, all_regs = [], []
    for r in reads:
        seq = np.ascontiguousarray(r, dtype=np.uint8)
        cv = L.mem_chain(opt, idx.contents.bwt, l_pac, len(seq), seq.ctypes.data_as(C.c_void_p))
        n = L.mem_chain_flt(opt, int(cv.n), cv.a)
        chains = []
        regs = AlnregV(0, 0, None)
        for ci in range(n):
            c = cv.a[ci]
            sd = np.zeros(c.n, dtype=kswlib.SEED)
            for k in range(c.n):
                sd[k] = (c.seeds[k].rbeg, c.seeds[k].qbeg, c.seeds[k].len)
            chains.append(sd)
            if run_chain2aln:
                L.mem_chain2aln(opt, l_pac, idx.contents.pac, len(seq), seq.ctypes.data_as(C.c_void_p),
                                C.byref(cv.a[ci]), C.byref(regs))
        out = np.zeros(regs.n, dtype=kswlib.ALNREG)
        if regs.n:
            C.memmove(out.ctypes.data, regs.a, regs.n * kswlib.ALNREG.itemsize)
        if regs.a:
            _libc.free(regs.a)
        for ci in range(n):  # mem_chain_flt already freed the seeds of the chains it dropped (bwamem.c:362-368)
            _libc.free(C.cast(cv.a[ci].seeds, C.c_void_p))
        if cv.a:
            _libc.free(C.cast(cv.a, C.c_void_p))
        all_chains.append(chains)
        all_regs.append(out)
    return all_chains, all_regs


def pac_of(idx):
    l_pac = idx.contents.bns.contents.l_pac
    n = l_pac // 4 + 1
    return int(l_pac), np.ctypeslib.as_array(idx.contents.pac, shape=(n,)).copy()


class MemAln(C.Structure):  # mem_aln_t, bwamem.h:72-82
    _fields_ = [("pos", C.c_int64), ("rid", C.c_int), ("flag", C.c_int), ("bits", C.c_uint32), ("n_cigar", C.c_int),
                ("cigar", C.POINTER(C.c_uint32)), ("score", C.c_int), ("sub", C.c_int)]


def ref_reg2aln(idx, opt, read, reg):
    """The reference's own mem_reg2aln (bwamem.c:1164-1236) on one region.  Returns
    (n_cigar, cigar words incl. clipping, NM, MD string, is_rev, pos)."""
    L = lib()
    L.mem_reg2aln.restype = MemAln
    L.mem_reg2aln.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    seq = np.ascontiguousarray(read, dtype=np.uint8)
    ar = np.ascontiguousarray(np.asarray(reg, dtype=kswlib.ALNREG).reshape(()))
    a = L.mem_reg2aln(opt, idx.contents.bns, idx.contents.pac, len(seq), seq.ctypes.data_as(C.c_void_p),
                      ar.ctypes.data_as(C.c_void_p))
    words = np.array([a.cigar[i] for i in range(a.n_cigar)], dtype=np.uint32)
    md = C.string_at(C.cast(a.cigar, C.c_void_p).value + 4 * a.n_cigar) if a.n_cigar or a.cigar else b""
    out = (a.n_cigar, words, (a.bits >> 9) & 0x7fffff, md, a.bits & 1, a.pos)
    if a.cigar:
        _libc.free(C.cast(a.cigar, C.c_void_p))
    return out


# ---- paired-end pieces: phase 1 per read, insert-size statistics, the reference's own mate rescue -------------------
def ref_align_reads(idx, opt, reads):
    """mem_align1_core (reference bwamem.c:1122) per read -> list of ALNREG arrays."""
    L = lib()
    L.mem_align1_core.restype = AlnregV
    L.mem_align1_core.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    out = []
    for r in reads:
        seq = np.ascontiguousarray(r, dtype=np.uint8).copy()  # mutated to codes in place (already codes here)
        v = L.mem_align1_core(opt, idx.contents.bwt, idx.contents.bns, idx.contents.pac, len(seq), seq.ctypes.data_as(C.c_void_p))
        a = np.zeros(v.n, dtype=kswlib.ALNREG)
        if v.n:
            C.memmove(a.ctypes.data, v.a, v.n * kswlib.ALNREG.itemsize)
        if v.a:
            _libc.free(v.a)
        out.append(a)
    return out


def ref_pestat(idx, opt, regs):
    """mem_pestat (reference bwamem_pair.c:46-107) over all pairs -> PESTAT[4]."""
    L = lib()
    l_pac = idx.contents.bns.contents.l_pac
    c_regs = kswlib.regs_to_c(regs)
    pes = np.zeros(4, dtype=kswlib.PESTAT)
    L.mem_pestat.restype = None
    L.mem_pestat(opt, C.c_int64(l_pac), C.c_int(len(regs)), c_regs, pes.ctypes.data_as(C.c_void_p))
    kswlib.regs_from_c(c_regs)
    return pes


def ref_dedup_fn(opt):
    """The reference's mem_sort_and_dedup(n, a, opt->mask_level_redun) (bwamem.c:395) behind the bmh_dedup_fn shape."""
    L = lib()
    L.mem_sort_and_dedup.restype = C.c_int
    L.mem_sort_and_dedup.argtypes = [C.c_int, C.c_void_p, C.c_float]
    lvl = float(opt.contents.mask_level_redun)
    return kswlib.DEDUP_FN(lambda user, n, a: L.mem_sort_and_dedup(n, a, lvl))


def ref_matesw_pairs(idx, opt, pes, reads, regs):
    """The mate-rescue block of mem_sam_pe (reference bwamem_pair.c:251-263), driven from here with the reference's
    OWN mem_matesw.  Returns (regs after rescue, n per pair)."""
    L = lib()
    l_pac = idx.contents.bns.contents.l_pac
    L.mem_matesw.restype = C.c_int
    L.mem_matesw.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    pes = np.ascontiguousarray(pes, dtype=kswlib.PESTAT)
    c_regs = kswlib.regs_to_c(regs)
    ns = []
    o = opt.contents
    for k in range(len(reads) // 2):
        a = [np.zeros(0, kswlib.ALNREG), np.zeros(0, kswlib.ALNREG)]
        for i in range(2):
            v = c_regs[2 * k + i]
            cur = np.zeros(v.n, dtype=kswlib.ALNREG)
            if v.n:
                C.memmove(cur.ctypes.data, v.a, v.n * kswlib.ALNREG.itemsize)
            a[i] = cur[cur["score"] >= cur["score"][0] - o.pen_unpaired].copy() if len(cur) else cur
        n = 0
        for i in range(2):
            mate = np.ascontiguousarray(reads[2 * k + (1 - i)], dtype=np.uint8)
            for j in range(min(len(a[i]), o.max_matesw)):
                hit = np.ascontiguousarray(a[i][j:j + 1])
                n += L.mem_matesw(opt, l_pac, idx.contents.pac, pes.ctypes.data_as(C.c_void_p), hit.ctypes.data_as(C.c_void_p),
                                  len(mate), mate.ctypes.data_as(C.c_void_p), C.byref(c_regs, (2 * k + 1 - i) * C.sizeof(kswlib.CAlnregV)))
        ns.append(n)
    return kswlib.regs_from_c(c_regs), ns


# ---- FM-index: the reference's own bwt_t and query functions (bwt.h:45-57, bwt.c) -----------------------------------
class BwtT(C.Structure):
    _fields_ = [("primary", C.c_uint64), ("L2", C.c_uint64 * 5), ("seq_len", C.c_uint64), ("bwt_size", C.c_uint64),
                ("bwt", C.POINTER(C.c_uint32)), ("cnt_table", C.c_uint32 * 256), ("sa_intv", C.c_int), ("n_sa", C.c_uint64),
                ("sa", C.POINTER(C.c_uint64))]


class BwtIntvV(C.Structure):  # bwtintv_v, bwt.h:63
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("a", C.c_void_p)]


def bwt_arrays(idx):
    """(primary, L2[5], seq_len, bwt words, sa_intv, sa) copied out of the loaded index."""
    b = C.cast(idx.contents.bwt, C.POINTER(BwtT)).contents
    words = np.ctypeslib.as_array(b.bwt, shape=(b.bwt_size,)).copy()
    sa = np.ctypeslib.as_array(b.sa, shape=(b.n_sa,)).copy()
    return int(b.primary), [int(x) for x in b.L2], int(b.seq_len), words, int(b.sa_intv), sa


def ref_smem1(idx, read, x, min_intv):
    """The reference's bwt_smem1 (bwt.c:288) -> (ret, SMEM_INTV[])."""
    L = lib()
    L.bwt_smem1.restype = C.c_int
    L.bwt_smem1.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    read = np.ascontiguousarray(read, dtype=np.uint8)
    mem = BwtIntvV(0, 0, None)
    ret = L.bwt_smem1(idx.contents.bwt, len(read), read.ctypes.data_as(C.c_void_p), x, min_intv, C.byref(mem), None)
    out = np.zeros(mem.n, dtype=kswlib.SMEM_INTV)
    if mem.n:
        C.memmove(out.ctypes.data, mem.a, mem.n * 32)
    if mem.a:
        _libc.free(mem.a)
    return ret, out


def ref_smem_iter(idx, opt, read):
    """smem_next2 (bwamem.c:118) iterated as mem_insert_seed does (bwamem.c:208-214): list of (start after the
    iteration, merged SMEM_INTV[])."""
    L = lib()
    L.smem_itr_init.restype = C.c_void_p
    L.smem_itr_init.argtypes = [C.c_void_p]
    L.smem_set_query.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.smem_next2.restype = C.POINTER(BwtIntvV)
    L.smem_next2.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.smem_itr_destroy.argtypes = [C.c_void_p]
    read = np.ascontiguousarray(read, dtype=np.uint8)
    o = opt.contents
    split_len = min(int(o.min_seed_len * o.split_factor + .499), len(read))
    itr = L.smem_itr_init(idx.contents.bwt)
    L.smem_set_query(itr, len(read), read.ctypes.data_as(C.c_void_p))
    out = []
    while True:
        a = L.smem_next2(itr, split_len, o.split_width, 2 if o.flag & 0x40 else 1)
        if not a:
            break
        v = np.zeros(a.contents.n, dtype=kswlib.SMEM_INTV)
        if a.contents.n:
            C.memmove(v.ctypes.data, a.contents.a, a.contents.n * 32)
        out.append(v)
    L.smem_itr_destroy(itr)
    return out


def ref_sa(idx, ks):
    L = lib()
    L.bwt_sa.restype = C.c_uint64
    L.bwt_sa.argtypes = [C.c_void_p, C.c_uint64]
    return np.array([L.bwt_sa(idx.contents.bwt, int(k)) for k in ks], dtype=np.uint64)


def smem_opt_of(opt, read_len=None):
    o = opt.contents
    so = np.zeros((), dtype=kswlib.SMEM_OPT)
    so["min_seed_len"], so["split_len"] = o.min_seed_len, int(o.min_seed_len * o.split_factor + .499)
    so["split_width"], so["start_width"] = o.split_width, 2 if o.flag & 0x40 else 1
    return so
