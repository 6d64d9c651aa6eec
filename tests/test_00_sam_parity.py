"""Whole-pipeline DUT/REF parity -- the reference author's own acceptance test (pipeline.sh vs
pipeline_ref.sh: SAM of the fork == SAM of stock).

REF = the reference compiled by oracle/Makefile (oracle/_ref/bwa), untouched.
DUT = the same binary with libbwamem_hip_dropin.so LD_PRELOADed.  Everything between read parsing and printing is then this
      library: phase 1 goes through the fork's batching seam mem_align1_core_batched -> the batch's FM-index queries on the GPU
      (bmh_seed_batch: SMEMs + suffix-array look-ups) -> the library's own chaining (bmh_chain_reads: mem_chain / mem_chain_flt
      restated, asserted below from the shim's log) -> bmh_chains2regs_batch (fused per-seed extension records on the GPU,
      mem_chain2aln_short's ksw_align2 calls as ONE bmh_sw_batch) -> bmh_sort_and_dedup; mem_process_seqs itself is taken over:
      insert-size statistics (bmh_pestat), the whole chunk's mate rescue (bmh_matesw_batch) and phase 2 (bmh_sam_batch: primary
      marking, pairing, mapQ, the global alignments of exactly the printed regions as GPU batches, SAM text).  Nothing of the
      reference's bwamem.c / bwamem_pair.c / bwt.c / ksw.c runs inside mem_process_seqs; the host program keeps FASTQ parsing, its
      thread pool, clocks and printing.
SAM must be byte-identical except the @PG header line.  Runs first in the session (file name) so
the parent process is GPU-clean when it starts the child processes."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import kswgen
import kswlib
import reflib
from __graft_entry__ import load_package

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not reflib.have_ref_bwa(), reason="oracle/_ref not built")]


def _sim_reads(rng, ref, n, L, hard, pair=False, rescue=0.0):
    r1, r2 = [], []
    for _ in range(n):
        ins = int(rng.integers(250, 450)) if pair else L
        pos = int(rng.integers(0, len(ref) - ins - 60))
        frag = ref[pos:pos + ins + 40]
        sub, ind, mx = (0.03, 0.008, 10) if hard else (0.02, 0.0025, 1)
        a = kswgen.mutate(rng, frag[:L + 30], sub, ind, ind, mx)[:L]
        if hard and rng.random() < 0.25:
            cut = int(rng.integers(40, L - 20))
            p2 = int(rng.integers(0, len(ref) - L))
            a = np.concatenate([a[:cut], ref[p2:p2 + L - cut]])
        a = a.copy()
        if hard:
            a[rng.random(len(a)) < 0.01] = 4
        if pair:
            if rng.random() < rescue:  # too many errors for a 19-mer seed: only mate rescue (ksw_align2) can place it
                b = kswgen.mutate(rng, frag[ins - L:ins + 30], 0.10, 0.004, 0.004, 2)[:L]
            else:
                b = kswgen.mutate(rng, frag[ins - L:ins + 30], sub, ind, ind, mx)[:L]
            b = (3 - b[::-1]).astype(np.uint8)
            r1.append(a), r2.append(b)
        else:
            if rng.random() < 0.5:
                a = np.where(a[::-1] > 3, 4, 3 - a[::-1]).astype(np.uint8)
            r1.append(a)
    return r1, r2


@pytest.fixture(scope="module")
def genome():
    rng = np.random.default_rng(424242)
    tmp = tempfile.mkdtemp(prefix="bmh_sam_")
    ref = kswgen.rand_seq(rng, 400000)
    for _ in range(30):  # planted diverged repeats -> multi-chain reads, secondary hits, mapQ ties
        a, b, L = int(rng.integers(0, 380000)), int(rng.integers(0, 380000)), int(rng.integers(200, 800))
        ref[b:b + L] = kswgen.mutate(rng, ref[a:a + L + 20], 0.02, 0.002, 0.002, 2)[:L]
    fa = os.path.join(tmp, "ref.fa")
    reflib.write_fasta(fa, "synth", ref)
    reflib.build_index(fa)
    return rng, tmp, fa, ref


def _run(fa, fqs, out, extra, preload, more_env=None):
    env = dict(os.environ)
    if preload:
        env["LD_PRELOAD"] = load_package().DROPIN_PATH
        env["BMH_VERBOSE"] = "1"
        env.update(more_env or {})
    with open(out, "w") as f:
        r = subprocess.run([reflib.REF_BWA, "mem", "-v", "1"] + extra + [fa] + fqs, stdout=f, stderr=subprocess.PIPE,
                           env=env, timeout=600)
    assert r.returncode == 0, f"bwa mem {extra} (preload={preload}) failed with {r.returncode}: {r.stderr.decode()[-2000:]}"
    _run.last_stderr = r.stderr.decode()
    return [l for l in open(out) if not l.startswith("@PG")]


@pytest.mark.parametrize("extra", [["-t", "4", "-b", "512"], ["-t", "2", "-b", "64", "-w", "10", "-d", "30"],
                                   ["-t", "3", "-b", "1000", "-A", "2", "-B", "6", "-O", "8,6", "-E", "2,3"],
                                   # seeding options: shorter seeds, re-seeding of every long match, few occurrences
                                   ["-t", "4", "-b", "300", "-k", "14", "-r", "1.0", "-c", "20"]])
def test_se_sam_identical(genome, extra):
    rng, tmp, fa, ref = genome
    reads = _sim_reads(rng, ref, 1500, 150, False)[0] + _sim_reads(rng, ref, 700, 250, True)[0] + \
        _sim_reads(rng, ref, 300, 101, True)[0]
    fq = os.path.join(tmp, "se.fq")
    reflib.write_fastq(fq, reads)
    ref_sam = _run(fa, [fq], os.path.join(tmp, "ref.sam"), extra, False)
    dut_sam = _run(fa, [fq], os.path.join(tmp, "dut.sam"), extra, True, {"BMH_BATCH_EXACT": "1"})  # batches of exactly -b reads
    assert len(ref_sam) > len(reads)
    assert ref_sam == dut_sam
    # phase 2 (primary marking, global alignments as GPU batches, SAM text) was the library's own bmh_sam_batch, not the
    # reference's worker2: the shim says so per chunk
    import re
    assert re.search(r"phase 2 \(marking, pairing, global alignments, SAM\)", _run.last_stderr)
    # ... and phase 1's chaining the library's own bmh_chain_reads over the GPU's SMEM / suffix-array batches
    m = re.findall(r"phase 1 so far: (\d+) chains from bmh_chain_reads, (\d+) seeds extended", _run.last_stderr)
    assert m and int(m[-1][0]) >= len(reads) // 2 and int(m[-1][1]) >= len(reads) // 2


def test_pe_sam_identical(genome):
    rng, tmp, fa, ref = genome
    r1, r2 = _sim_reads(rng, ref, 1200, 150, False, pair=True)
    f1, f2 = os.path.join(tmp, "pe_1.fq"), os.path.join(tmp, "pe_2.fq")
    reflib.write_fastq(f1, r1, "p")
    reflib.write_fastq(f2, r2, "p")
    extra = ["-t", "4", "-b", "400"]
    ref_sam = _run(fa, [f1, f2], os.path.join(tmp, "ref_pe.sam"), extra, False)
    dut_sam = _run(fa, [f1, f2], os.path.join(tmp, "dut_pe.sam"), extra, True, {"BMH_BATCH_EXACT": "1"})
    assert len(ref_sam) >= 2400
    assert ref_sam == dut_sam
    # ... and with the batch size evened out over the chunk (the shim's default) and GPU waits that spin
    assert ref_sam == _run(fa, [f1, f2], os.path.join(tmp, "dut_pe2.sam"), extra, True, {"BMH_WAIT": "spin"})


def test_pe_mate_rescue_sam_identical(genome):
    """Pairs whose second mate cannot be seeded: mem_matesw (reference bwamem_pair.c:109-175) places it with
    ksw_align2, which the preload routes to the GPU Smith-Waterman kernels."""
    rng, tmp, fa, ref = genome
    r1, r2 = _sim_reads(rng, ref, 900, 150, False, pair=True, rescue=0.5)
    h1, h2 = _sim_reads(rng, ref, 300, 125, True, pair=True, rescue=0.5)
    f1, f2 = os.path.join(tmp, "mr_1.fq"), os.path.join(tmp, "mr_2.fq")
    reflib.write_fastq(f1, r1 + h1, "m")
    reflib.write_fastq(f2, r2 + h2, "m")
    extra = ["-t", "4", "-b", "300"]
    ref_sam = _run(fa, [f1, f2], os.path.join(tmp, "ref_mr.sam"), extra, False)
    dut_sam = _run(fa, [f1, f2], os.path.join(tmp, "dut_mr.sam"), extra, True)
    assert len(ref_sam) >= 2400
    # the scenario really exercises rescue: a good share of second mates is placed although it had no seed hit
    placed = sum(1 for l in ref_sam if not l.startswith("@") and int(l.split("\t")[1]) & 0x80 and not int(l.split("\t")[1]) & 0x4)
    assert placed > 900
    assert ref_sam == dut_sam
    # ... and the DUT did it through the batched seam (mem_process_seqs -> bmh_matesw_batch), not call by call
    import re
    m = re.findall(r"mate rescue: (\d+) pairs, (\d+) ksw_align2 calls in (\d+) GPU rounds, (\d+) pool bytes", _run.last_stderr)
    assert m and sum(int(x[1]) for x in m) > 100 and all(int(x[2]) <= 12 for x in m)
    assert all(int(x[3]) <= int(x[0]) * 2 * 151 + 64 for x in m), "with the reference resident only the reads are shipped"


def test_pe_250bp_mate_rescue_sam_identical(genome):
    """2 x 250 bp: mem_matesw calls ksw_align2 in WORD mode (l_ms * a >= 250, bwamem_pair.c:147) -- on small batches the
    320-column one-wave-per-task kernel, with BMH_SW_WAVE=0 the 256-column register kernels."""
    rng, tmp, fa, ref = genome
    r1, r2 = [], []
    for _ in range(700):
        ins = int(rng.integers(400, 700))
        pos = int(rng.integers(0, len(ref) - ins - 60))
        frag = ref[pos:pos + ins + 40]
        a = kswgen.mutate(rng, frag[:280], 0.02, 0.0025, 0.0025, 1)[:250]
        b = kswgen.mutate(rng, frag[ins - 250:ins + 30], 0.02, 0.004, 0.004, 2)[:250].copy()
        if rng.random() < 0.5:  # a mismatch every 15 bases: no 19-mer seed anywhere, yet a local alignment of 170+ for mem_matesw
            at = np.arange(int(rng.integers(0, 15)), 250, 15)
            b[at] = (b[at] + rng.integers(1, 4, len(at))) & 3
        r1.append(a.copy()), r2.append((3 - b[::-1]).astype(np.uint8))
    f1, f2 = os.path.join(tmp, "l250_1.fq"), os.path.join(tmp, "l250_2.fq")
    reflib.write_fastq(f1, r1, "w")
    reflib.write_fastq(f2, r2, "w")
    extra = ["-t", "4", "-b", "200"]
    ref_sam = _run(fa, [f1, f2], os.path.join(tmp, "ref_250.sam"), extra, False)
    assert len(ref_sam) >= 1400
    assert ref_sam == _run(fa, [f1, f2], os.path.join(tmp, "dut_250.sam"), extra, True)
    import re
    m = re.findall(r"mate rescue: (\d+) pairs, (\d+) ksw_align2 calls", _run.last_stderr)
    assert m and sum(int(x[1]) for x in m) > 100
    assert ref_sam == _run(fa, [f1, f2], os.path.join(tmp, "dut_250l.sam"), extra, True, {"BMH_SW_WAVE": "0"})


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_fuzz_ragged_reads_sam_identical(genome, seed):
    """Ragged input: lengths from below min_seed_len to 260, N runs, mates of different length, error rates from 0 to 15 %,
    chimeric reads -- SE and PE, a different mix per seed."""
    _, tmp, fa, ref = genome
    rng = np.random.default_rng(seed)

    def one(L, rate):
        pos = int(rng.integers(0, len(ref) - 700))
        ins = int(rng.integers(max(L, 60), 600))
        frag = ref[pos:pos + ins + 40]
        a = kswgen.mutate(rng, frag[:L + 30], rate, rate / 8, rate / 8, 3)[:L].copy()
        if rng.random() < 0.15 and L > 60:  # chimera
            cut = int(rng.integers(25, L - 25))
            p2 = int(rng.integers(0, len(ref) - L))
            a = np.concatenate([a[:cut], ref[p2:p2 + L - cut]]).copy()
        if rng.random() < 0.2:
            k = int(rng.integers(0, max(1, L - 5)))
            a[k:k + int(rng.integers(1, 12))] = 4
        return a, frag, ins

    lens = [12, 18, 19, 20, 35, 50, 76, 100, 101, 125, 150, 151, 200, 250, 260]
    se, p1, p2 = [], [], []
    for _ in range(700):
        a, _, _ = one(int(rng.choice(lens)), float(rng.choice([0.0, 0.01, 0.03, 0.08, 0.15])))
        se.append(a if rng.random() < 0.5 else np.where(a[::-1] > 3, 4, 3 - a[::-1]).astype(np.uint8))
    for _ in range(500):
        L1, L2 = int(rng.choice(lens)), int(rng.choice(lens))
        a, frag, ins = one(L1, float(rng.choice([0.0, 0.02, 0.06])))
        lo = max(0, ins - L2)
        b = kswgen.mutate(rng, frag[lo:lo + L2 + 30], float(rng.choice([0.01, 0.05, 0.12])), 0.004, 0.004, 2)[:L2]
        if len(b) < 1:
            continue
        p1.append(a), p2.append(np.where(b[::-1] > 3, 4, 3 - b[::-1]).astype(np.uint8))
    fq = os.path.join(tmp, f"fz{seed}.fq")
    reflib.write_fastq(fq, se, "z")
    extra = ["-t", "4", "-b", "173"]
    assert _run(fa, [fq], os.path.join(tmp, f"ref_fz{seed}.sam"), extra, False) == \
        _run(fa, [fq], os.path.join(tmp, f"dut_fz{seed}.sam"), extra, True, {"BMH_BATCH_EXACT": "1"})
    f1, f2 = os.path.join(tmp, f"fz{seed}_1.fq"), os.path.join(tmp, f"fz{seed}_2.fq")
    reflib.write_fastq(f1, p1, "y")
    reflib.write_fastq(f2, p2, "y")
    ref_sam = _run(fa, [f1, f2], os.path.join(tmp, f"ref_fzp{seed}.sam"), extra, False)
    assert len(ref_sam) >= 2 * len(p1)
    assert ref_sam == _run(fa, [f1, f2], os.path.join(tmp, f"dut_fzp{seed}.sam"), extra, True)


def test_multi_contig_reference_sam_identical():
    """Three contigs of different length, reads that hang over contig ends and reads with N: exercises bwa_fix_xref2
    (reference bwa.c:179) in front of the CIGAR batch and the rid/pos conversion behind it, SE and PE."""
    rng = np.random.default_rng(515151)
    tmp = tempfile.mkdtemp(prefix="bmh_mc_")
    contigs = [kswgen.rand_seq(rng, n) for n in (90000, 30011, 6007)]
    fa = os.path.join(tmp, "mc.fa")
    with open(fa, "w") as f:
        for k, c in enumerate(contigs):
            f.write(f">ctg{k} some description\n")
            s = "".join("ACGT"[b] for b in c)
            for i in range(0, len(s), 70):
                f.write(s[i:i + 70] + "\n")
    reflib.build_index(fa)
    whole = np.concatenate(contigs)
    reads = []
    for _ in range(900):
        L = int(rng.choice([100, 150, 151, 220]))
        pos = int(rng.integers(0, len(whole) - L))  # may straddle a contig boundary
        rd = kswgen.mutate(rng, whole[pos:pos + L + 10], 0.02, 0.003, 0.003, 3)[:L].copy()
        if rng.random() < 0.2:
            rd[rng.random(len(rd)) < 0.02] = 4
        if rng.random() < 0.5:
            rd = np.where(rd[::-1] > 3, 4, 3 - rd[::-1]).astype(np.uint8)
        reads.append(rd)
    for b in (0, 90000 - 70, 90000 - 20, 120011 - 75, len(whole) - 150):  # deliberately across / at the ends
        reads.append(whole[b:b + 150].copy())
    fq = os.path.join(tmp, "mc.fq")
    reflib.write_fastq(fq, reads)
    extra = ["-t", "3", "-b", "200"]
    ref_sam = _run(fa, [fq], os.path.join(tmp, "ref.sam"), extra, False)
    dut_sam = _run(fa, [fq], os.path.join(tmp, "dut.sam"), extra, True)
    assert len(ref_sam) > len(reads) and ref_sam == dut_sam
    r1, r2 = _sim_reads(rng, whole, 500, 125, False, pair=True, rescue=0.4)
    f1, f2 = os.path.join(tmp, "mc_1.fq"), os.path.join(tmp, "mc_2.fq")
    reflib.write_fastq(f1, r1, "q")
    reflib.write_fastq(f2, r2, "q")
    ref_sam = _run(fa, [f1, f2], os.path.join(tmp, "ref_pe.sam"), extra, False)
    dut_sam = _run(fa, [f1, f2], os.path.join(tmp, "dut_pe.sam"), extra, True)
    assert len(ref_sam) >= 1000 and ref_sam == dut_sam


@pytest.mark.parametrize("extra", [["-t", "1", "-b", "100000", "-a", "-M"],      # one thread, one batch, all alignments, -M marking
                                   ["-t", "4", "-b", "77", "-S"],                 # mate rescue switched off in the reference
                                   ["-t", "3", "-b", "256", "-P", "-T", "45"]])   # no pairing, higher output threshold
def test_pe_option_matrix_sam_identical(genome, extra):
    """Option combinations that change WHICH regions reach mem_reg2aln / mem_matesw (reference fastmap.c:56-100):
    the chunk-wide batches are supersets and the tables must still serve exactly what the reference asks for."""
    rng, tmp, fa, ref = genome
    r1, r2 = _sim_reads(rng, ref, 400, 150, True, pair=True, rescue=0.3)
    s1 = [x[:int(rng.integers(15, 150))] for x in r1[:40]]  # some mates shorter than a seed, some barely longer
    f1, f2 = os.path.join(tmp, "om_1.fq"), os.path.join(tmp, "om_2.fq")
    reflib.write_fastq(f1, s1 + r1[40:], "o")
    reflib.write_fastq(f2, r2, "o")
    ref_sam = _run(fa, [f1, f2], os.path.join(tmp, "ref_om.sam"), extra, False)
    dut_sam = _run(fa, [f1, f2], os.path.join(tmp, "dut_om.sam"), extra, True)
    assert len(ref_sam) >= 800 and ref_sam == dut_sam


def test_pe_sam_identical_with_a_device_list(genome):
    """$BMH_DEVICES spreads the shim's contexts over several GPUs (each gets its own resident reference and index).  The
    box has one GPU, so the list names it twice: the contexts alternate between two entries that happen to be one device."""
    rng, tmp, fa, ref = genome
    r1, r2 = _sim_reads(rng, ref, 600, 150, False, pair=True, rescue=0.3)
    fq1, fq2 = os.path.join(tmp, "dl_1.fq"), os.path.join(tmp, "dl_2.fq")
    reflib.write_fastq(fq1, r1, "p")
    reflib.write_fastq(fq2, r2, "p")
    extra = ["-t", "4", "-b", "256"]
    ref_sam = _run(fa, [fq1, fq2], os.path.join(tmp, "ref_dl.sam"), extra, False)
    dut_sam = _run(fa, [fq1, fq2], os.path.join(tmp, "dut_dl.sam"), extra, True, {"BMH_DEVICES": "0,0"})
    assert len(ref_sam) >= 1200 and ref_sam == dut_sam


def test_pe_sam_identical_over_all_visible_devices(genome):
    """$BMH_DEVICES naming EVERY visible GPU (one on this pool, eight on a node): the shim's contexts land on all of them in
    turn, each device holds its own resident reference and index, no exchange between them -- and the SAM is the reference's."""
    import torch
    ndev = torch.cuda.device_count()
    rng, tmp, fa, ref = genome
    r1, r2 = _sim_reads(rng, ref, 900, 150, False, pair=True, rescue=0.3)
    fq1, fq2 = os.path.join(tmp, "ad_1.fq"), os.path.join(tmp, "ad_2.fq")
    reflib.write_fastq(fq1, r1, "p")
    reflib.write_fastq(fq2, r2, "p")
    extra = ["-t", str(max(4, 2 * ndev)), "-b", "128"]
    ref_sam = _run(fa, [fq1, fq2], os.path.join(tmp, "ref_ad.sam"), extra, False)
    dut_sam = _run(fa, [fq1, fq2], os.path.join(tmp, "dut_ad.sam"), extra, True, {"BMH_DEVICES": ",".join(str(d) for d in range(ndev))})
    assert len(ref_sam) >= 1800 and ref_sam == dut_sam
