"""bmh_index_load: the index files `bwa index` writes, read without the reference's code, must equal what the
reference's own loader (bwa_idx_load, reference bwa.c:270-300) holds in memory."""
import ctypes as C
import os

import numpy as np
import pytest

import kswgen
import kswlib
import reflib
from __graft_entry__ import load_package

pytestmark = pytest.mark.skipif(not reflib.have_ref_bwa(), reason="oracle/_ref not built (no /root/reference here)")


class CIndex(C.Structure):  # bmh_index_t
    _fields_ = [("bwt", kswlib.CBwt), ("l_pac", C.c_int64), ("pac", C.POINTER(C.c_uint8)), ("n_seqs", C.c_int32),
                ("names", C.POINTER(C.c_char_p)), ("offsets", C.POINTER(C.c_int64)), ("lens", C.POINTER(C.c_int32)),
                ("n_holes", C.c_int32), ("hole_offsets", C.POINTER(C.c_int64)), ("hole_lens", C.POINTER(C.c_int32)),
                ("hole_chars", C.POINTER(C.c_char))]


def test_index_files_load_like_the_reference(tmp_path):
    rng = np.random.default_rng(211)
    contigs = [kswgen.rand_seq(rng, n) for n in (30011, 5003, 977)]
    fa = str(tmp_path / "ix.fa")
    with open(fa, "w") as f:
        for k, c in enumerate(contigs):
            f.write(f">c{k}" + (" with a comment\n" if k == 1 else "\n"))
            txt = "".join("ACGT"[b] for b in c)
            if k == 0:  # two runs of ambiguous bases: the .amb file records them as holes (bntseq.c:136-150)
                txt = txt[:1000] + "N" * 37 + txt[1037:20000] + "RRR" + txt[20003:]
            f.write(txt + "\n")
    reflib.build_index(fa)
    idx = reflib.lib().bwa_idx_load(fa.encode(), 7)
    prim, L2, sl, words, sai, sa = reflib.bwt_arrays(idx)
    l_pac, pac = reflib.pac_of(idx)
    L = load_package().lib()
    L.bmh_index_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(CIndex))]
    L.bmh_index_free.argtypes = [C.POINTER(CIndex)]
    px = C.POINTER(CIndex)()
    assert L.bmh_index_load(fa.encode(), C.byref(px)) == 0
    ix = px.contents
    assert (ix.bwt.primary, [ix.bwt.L2[i] for i in range(5)], ix.bwt.seq_len, ix.bwt.bwt_size, ix.bwt.sa_intv, ix.bwt.n_sa) == \
        (prim, L2, sl, len(words), sai, len(sa))
    assert np.array_equal(np.ctypeslib.as_array(C.cast(ix.bwt.bwt, C.POINTER(C.c_uint32)), shape=(len(words),)), words)
    assert np.array_equal(np.ctypeslib.as_array(C.cast(ix.bwt.sa, C.POINTER(C.c_uint64)), shape=(len(sa),)), sa)
    assert ix.l_pac == l_pac and np.array_equal(np.ctypeslib.as_array(ix.pac, shape=(l_pac // 4 + 1,)), pac)
    assert ix.n_seqs == 3 and [ix.names[i] for i in range(3)] == [b"c0", b"c1", b"c2"]
    assert [ix.offsets[i] for i in range(3)] == [0, 30011, 35014] and [ix.lens[i] for i in range(3)] == [30011, 5003, 977]
    # holes: as the reference's own loader holds them (bntamb1_t: offset, len, letter)
    class Amb(C.Structure):
        _fields_ = [("offset", C.c_int64), ("len", C.c_int32), ("amb", C.c_char)]

    class Bns(C.Structure):  # bntseq_t, bntseq.h:53-61
        _fields_ = [("l_pac", C.c_int64), ("n_seqs", C.c_int32), ("seed", C.c_uint32), ("anns", C.c_void_p), ("n_holes", C.c_int32),
                    ("ambs", C.POINTER(Amb))]
    bns = C.cast(idx.contents.bns, C.POINTER(Bns)).contents
    assert ix.n_holes == bns.n_holes == 2
    assert [(ix.hole_offsets[i], ix.hole_lens[i], ix.hole_chars[i]) for i in range(2)] == [(bns.ambs[i].offset, bns.ambs[i].len, bns.ambs[i].amb) for i in range(2)] \
        == [(1000, 37, b"N"), (20000, 3, b"R")]
    L.bmh_index_free(px)
    assert L.bmh_index_load(os.path.join(str(tmp_path), "missing").encode(), C.byref(px)) != 0
