#!/usr/bin/env python3
"""Writes the synthetic paired-end input of bench.py's whole-pipeline baseline (genome FASTA indexed by oracle/_ref/bwa,
two FASTQ files) into a directory, for profiling the DUT by hand:  python tools/make_pipeline_input.py DIR [n_reads] [genome_bp]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import kswgen  # noqa: E402
import reflib  # noqa: E402

out, n_reads, G = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1_600_000, int(sys.argv[3]) if len(sys.argv) > 3 else 4_600_000
os.makedirs(out, exist_ok=True)
rng = np.random.default_rng(20261101)
ref = kswgen.rand_seq(rng, G)
fa = os.path.join(out, "ref.fa")
reflib.write_fasta(fa, "synth", ref)
reflib.build_index(fa)
n_pairs, L = n_reads // 2, 150
ins = rng.integers(250, 450, size=n_pairs)
pos = rng.integers(0, len(ref) - 520, size=n_pairs)
idx = np.arange(L)[None, :]
a, b = ref[pos[:, None] + idx], ref[(pos + ins - L)[:, None] + idx]
rate_b = np.where(rng.random(n_pairs) < 0.3, 0.12, 0.02)[:, None]
for arr, rate in ((a, 0.02), (b, rate_b)):
    sub = rng.random(arr.shape) < rate
    arr[sub] = (arr[sub] + rng.integers(1, 4, size=int(sub.sum()))) & 3
b = 3 - b[:, ::-1]
bench._fastq_fixed(os.path.join(out, "r1.fq"), a, "p")
bench._fastq_fixed(os.path.join(out, "r2.fq"), b, "p")
print(out)
