"""mem_align1_core (reference bwamem.c:1122-1149), the per-read entry of the call surface (mem_align1 :1151, example.c:44), as exported
by libbwamem_hip_dropin.so with the reference's exact signature: same regions as the reference's own function on the same reads.
Both libraries are loaded into ONE child process (the shim takes its base-code table and clocks from the host program, here the
reference library loaded RTLD_GLOBAL); the child prints one line per read."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import kswgen
import reflib
from __graft_entry__ import load_package

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not reflib.have_ref_bwa(), reason="oracle/_ref not built")]

CHILD = r'''
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, sys.argv[4]); sys.path.insert(0, os.path.dirname(sys.argv[4]))
import reflib, kswlib
ref = C.CDLL(reflib.REF_LIB, mode=C.RTLD_GLOBAL)          # the host program: nst_nt4_table, realtime, kt_for, bwa_verbose ...
dut = C.CDLL(sys.argv[1])                                   # the shim (its own mem_align1_core, found through ITS handle)
ref.mem_opt_init.restype = C.POINTER(reflib.MemOpt)
ref.bwa_idx_load.restype = C.POINTER(reflib.BwaIdx); ref.bwa_idx_load.argtypes = [C.c_char_p, C.c_int]
C.c_int.in_dll(ref, "bwa_verbose").value = 1
opt = ref.mem_opt_init()
for kv in sys.argv[5:]:
    k, v = kv.split("="); setattr(opt.contents, k, int(v))
ref.bwa_fill_scmat(opt.contents.a, opt.contents.b, opt.contents.mat)
idx = ref.bwa_idx_load(sys.argv[2].encode(), 7)
reads = np.load(sys.argv[3], allow_pickle=False)
lens = reads["lens"]; flat = reads["flat"]; off = np.concatenate([[0], np.cumsum(lens)])
bad = 0
for f in (ref, dut):
    f.mem_align1_core.restype = reflib.AlnregV
    f.mem_align1_core.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_void_p]
libc = C.CDLL(None); libc.free.argtypes = [C.c_void_p]
def run(f, seq):
    s = np.ascontiguousarray(seq).copy()
    v = f.mem_align1_core(opt, idx.contents.bwt, idx.contents.bns, idx.contents.pac, len(s), s.ctypes.data_as(C.c_void_p))
    a = np.zeros(v.n, dtype=kswlib.ALNREG)
    if v.n: C.memmove(a.ctypes.data, v.a, v.n * kswlib.ALNREG.itemsize)
    if v.a: libc.free(v.a)
    return a, s
n_regs = 0
for r in range(len(lens)):
    seq = flat[off[r]:off[r + 1]]
    a, s1 = run(ref, seq)
    b, s2 = run(dut, seq)
    n_regs += len(a)
    same = len(a) == len(b) and all((a[f] == b[f]).all() for f in ("rb", "re", "qb", "qe", "score", "truesc", "sub", "csub", "sub_n", "w", "seedcov", "secondary"))
    same = same and (s1 == s2).all() and s2.max() <= 4       # the read was converted to codes in place (bwamem.c:1128-1129)
    bad += not same
print("RESULT", len(lens), n_regs, bad)
'''


def test_mem_align1_core_matches_the_reference_function():
    rng = np.random.default_rng(99)
    tmp = tempfile.mkdtemp(prefix="bmh_a1_")
    ref = kswgen.rand_seq(rng, 200000)
    for _ in range(12):
        a, b, L = int(rng.integers(0, 190000)), int(rng.integers(0, 190000)), int(rng.integers(200, 600))
        ref[b:b + L] = kswgen.mutate(rng, ref[a:a + L + 20], 0.02, 0.002, 0.002, 2)[:L]
    fa = os.path.join(tmp, "ref.fa")
    reflib.write_fasta(fa, "synth", ref)
    reflib.build_index(fa)
    reads = []
    for _ in range(160):
        L = int(rng.choice([70, 101, 150, 250]))
        p = int(rng.integers(0, len(ref) - L - 40))
        r = kswgen.mutate(rng, ref[p:p + L + 30], 0.03, 0.006, 0.006, 6)[:L].copy()
        if rng.random() < 0.5:
            r = (3 - r[::-1]).astype(np.uint8)
        if rng.random() < 0.3:  # as letters: the function converts them itself
            r = np.frombuffer(b"ACGT", dtype=np.uint8)[r].copy()
            if rng.random() < 0.5:
                r[int(rng.integers(0, L))] = ord("N")
        reads.append(r.astype(np.uint8))
    npz = os.path.join(tmp, "reads.npz")
    np.savez(npz, lens=np.array([len(r) for r in reads]), flat=np.concatenate(reads))
    script = os.path.join(tmp, "child.py")
    open(script, "w").write(CHILD)
    tests = os.path.dirname(os.path.abspath(__file__))
    for extra in ([], ["w=12", "zdrop=40"]):
        r = subprocess.run([sys.executable, script, load_package().DROPIN_PATH, fa, npz, tests] + extra, capture_output=True, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        line = [l for l in r.stdout.decode().splitlines() if l.startswith("RESULT")][0].split()
        assert int(line[1]) == len(reads) and int(line[2]) > len(reads) // 2 and int(line[3]) == 0, line
