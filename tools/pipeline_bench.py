#!/usr/bin/env python3
"""Whole-pipeline DUT/REF timing on the GPU box (informational; bench.py is the contract metric).

REF = the reference `bwa mem` compiled by oracle/Makefile; DUT = the same binary with
libbwamem_hip_dropin.so preloaded (mem_process_seqs entirely on the library: seeding, chaining, extension, mate rescue,
post-processing, SAM text).  Round 1's tool, kept for single-end runs and planted repeats; bench.py's pipeline_baseline and
tools/pipeline_matrix.sh are what the numbers in DESIGN.md come from.
Reads/s are taken from the reference's own per-chunk line
  [M::mem_process_seqs] Processed N reads in X CPU sec, Y real sec      (reference bwamem.c:1320-1321)
which excludes index loading.  SAM equality is checked as well.
Usage: python tools/pipeline_bench.py [--reads 200000] [--genome 4600000] [--threads 16,64]
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kswgen  # noqa: E402
import reflib  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402


def sim_reads_fast(rng, ref, n, L):
    """Vectorised read simulator: 2 % substitutions, 0.25 % 1-bp deletions, 0.25 % 1-bp insertions, 50 % revcomp."""
    M = 24  # spare bases so that deletions never leave the read short
    pos = rng.integers(0, len(ref) - L - M - 8, size=n)
    idx = pos[:, None] + np.arange(L + M)[None, :]
    frag = ref[idx]
    out = np.empty((n, L), dtype=np.uint8)
    for k in range(n):
        f = frag[k]
        u = rng.random(L + M)
        keep = u >= 0.0025                      # deletions
        f = f[keep]
        ins = np.nonzero(rng.random(len(f)) < 0.0025)[0]
        if len(ins):
            f = np.insert(f, ins, rng.integers(0, 4, size=len(ins)))
        f = f[:L].copy()
        sub = rng.random(L) < 0.02
        f[sub] = (f[sub] + rng.integers(1, 4, size=int(sub.sum()))) & 3
        if rng.random() < 0.5:
            f = (3 - f[::-1])
        out[k] = f
    return out


def sim_pairs_fast(rng, ref, n, L, noisy_frac):
    """Read pairs (FR, insert 250-450); a fraction of second mates carries 12 % substitutions, so that only mate
    rescue (mem_matesw -> ksw_align2) can place most of them."""
    ins = rng.integers(250, 450, size=n)
    pos = rng.integers(0, len(ref) - 520, size=n)
    a = np.empty((n, L), dtype=np.uint8)
    b = np.empty((n, L), dtype=np.uint8)
    for k in range(n):
        f = ref[pos[k]: pos[k] + L].copy()
        sub = rng.random(L) < 0.02
        f[sub] = (f[sub] + rng.integers(1, 4, size=int(sub.sum()))) & 3
        g = ref[pos[k] + ins[k] - L: pos[k] + ins[k]].copy()
        sub = rng.random(L) < (0.12 if rng.random() < noisy_frac else 0.02)
        g[sub] = (g[sub] + rng.integers(1, 4, size=int(sub.sum()))) & 3
        a[k], b[k] = f, 3 - g[::-1]
    return a, b


def run(fa, fq, threads, batch, preload, out, ksw_dropin=True):
    env = dict(os.environ)
    if preload:
        env["LD_PRELOAD"] = load_package().DROPIN_PATH
        env["BMH_KSW_DROPIN"] = "1" if ksw_dropin else "0"
        env["BMH_VERBOSE"] = "1"
        env["BMH_SMEM_TRACE"] = "1"
        if os.environ.get("BMH_DRIVER_TRACE"):
            env["BMH_DRIVER_TRACE"] = "1"
    t0 = time.time()
    with open(out, "w") as f:
        p = subprocess.run([reflib.REF_BWA, "mem", "-t", str(threads), "-b", str(batch), fa] + (fq if isinstance(fq, list) else [fq]), stdout=f,
                           stderr=subprocess.PIPE, env=env, check=True, timeout=3000)
    wall = time.time() - t0
    reads = real = 0
    for m in re.finditer(r"Processed (\d+) reads in ([\d.]+) CPU sec, ([\d.]+) real sec", p.stderr.decode()):
        reads += int(m.group(1))
        real += float(m.group(3))
    shim = [l for l in p.stderr.decode().splitlines() if l.startswith("[bwamem_hip]")]
    drv = [l for l in shim if "bmh_chain2aln_batch" in l]
    cgr = [l for l in shim if "bmh_reg2cigar_batch" in l and "regions:" in l]
    shim = [l for l in shim if l not in drv and l not in cgr]
    return {"reads": reads, "shim": shim[:6] + drv[2:8] + cgr[:4] + shim[-8:], "process_seqs_real_s": real, "reads_per_s": reads / real if real else None, "wall_s": wall}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=200000)
    ap.add_argument("--genome", type=int, default=4600000)
    ap.add_argument("--threads", default="16,64")
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--pe", action="store_true", help="paired-end reads; the DUT then also batches mate rescue on the GPU")
    ap.add_argument("--noisy", type=float, default=0.3, help="--pe: fraction of second mates that need rescue")
    ap.add_argument("--repeats", type=int, default=0, help="plant this many diverged repeats (200-3000 bp) in the genome")
    ap.add_argument("--full", action="store_true", help="also time the per-call ksw_global2 GPU drop-in (slow by design)")
    a = ap.parse_args()
    rng = np.random.default_rng(20261007)
    tmp = tempfile.mkdtemp(prefix="bmh_pipe_")
    ref = kswgen.rand_seq(rng, a.genome)
    for _ in range(a.repeats):  # multi-copy sequence: seeds with many occurrences, secondary hits, mapQ ties, more rescue
        src, dst, L = int(rng.integers(0, a.genome - 4000)), int(rng.integers(0, a.genome - 4000)), int(rng.integers(200, 3000))
        ref[dst:dst + L] = kswgen.mutate(rng, ref[src:src + L + 40], float(rng.choice([0.0, 0.005, 0.02])), 0.001, 0.001, 2)[:L]
    fa, fq = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "reads.fq")
    reflib.write_fasta(fa, "synth", ref)
    t0 = time.time()
    reflib.build_index(fa)
    t_index = time.time() - t0
    if a.pe:
        m1, m2 = sim_pairs_fast(rng, ref, a.reads // 2, 150, a.noisy)
        fq2 = os.path.join(tmp, "reads_2.fq")
        reflib.write_fastq(fq, list(m1), "p")
        reflib.write_fastq(fq2, list(m2), "p")
        fq = [fq, fq2]
    else:
        reads = sim_reads_fast(rng, ref, a.reads, 150)
        reflib.write_fastq(fq, list(reads))
    res = {"genome_bp": a.genome, "repeats": a.repeats, "reads": a.reads, "paired": bool(a.pe), "index_s": t_index, "runs": []}
    # one untimed DUT run first: on a fresh box the first process to load the HIP runtime and the library's code objects
    # pays for reading them from disk (seconds), which has nothing to do with the pipeline
    run(fa, fq, 8, a.batch, True, os.path.join(tmp, "warm.sam"), ksw_dropin=False)
    for t in [int(x) for x in a.threads.split(",")]:
        r = run(fa, fq, t, a.batch, False, os.path.join(tmp, "ref.sam"))
        refsam = [l for l in open(os.path.join(tmp, "ref.sam")) if not l.startswith("@PG")]
        d1 = run(fa, fq, t, a.batch, True, os.path.join(tmp, "dut.sam"), ksw_dropin=False)
        same1 = refsam == [l for l in open(os.path.join(tmp, "dut.sam")) if not l.startswith("@PG")]
        d2 = None
        if a.full:
            d2 = run(fa, fq, t, a.batch, True, os.path.join(tmp, "dut.sam"), ksw_dropin=True)
            d2["sam_identical"] = refsam == [l for l in open(os.path.join(tmp, "dut.sam")) if not l.startswith("@PG")]
        res["runs"].append({"threads": t, "batch": a.batch, "ref": r, "dut_phase1_gpu": d1, "sam_identical": same1,
                            "dut_phase1_gpu_plus_percall_global": d2})
        print(json.dumps(res["runs"][-1]), flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
