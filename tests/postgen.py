"""Seeded generators of region vectors for the post-processing tests (shared by tools/make_postproc_fixture.py and the
tests): many ties in end coordinate, score and position, overlapping and nested regions, both strands."""
import numpy as np

import kswlib

# mem_opt_t fields varied across fixture groups (defaults: reference bwamem.c:45-75)
OPTION_SETS = [dict(), dict(a=2, b=8, o_del=12, e_del=2, o_ins=12, e_ins=2, T=60), dict(mask_level=0.3, mask_level_redun=0.8, mapQ_coef_len=0.0),
               dict(min_seed_len=25, pen_unpaired=9, max_ins=600)]


def region_vectors(rng, n_reads, l_pac, L=150):
    out = []
    for _ in range(n_reads):
        n = int(rng.choice([0, 1, 1, 2, 3, 5, 8, 14, 23, 40]))
        a = np.zeros(n, dtype=kswlib.ALNREG)
        if n:
            loci = rng.integers(1000, 2 * l_pac - 1000, size=max(1, n // 3))
            for k in range(n):
                base = int(loci[rng.integers(0, len(loci))]) + int(rng.choice([0, 0, 0, 1, 3, 20, 60]))
                qb = int(rng.choice([0, 0, 5, 30, 70]))
                qe = int(rng.choice([L, L, L - 4, 100, qb + 25]))
                if qe <= qb:
                    qe = qb + 20
                ln = qe - qb + int(rng.choice([0, 0, 0, 1, -1, 3]))
                rb = min(max(base, 0), 2 * l_pac - 400)
                if rb < l_pac < rb + ln + 5:
                    rb = l_pac + 10
                sc = int(rng.choice([qe - qb, qe - qb, qe - qb - 5, qe - qb - 10, 30, 45, 19]))
                a[k]["rb"], a[k]["re"], a[k]["qb"], a[k]["qe"] = rb, rb + max(ln, 5), qb, qe
                a[k]["score"], a[k]["truesc"] = max(sc, 1), max(sc, 1) + int(rng.choice([0, 0, 3]))
                a[k]["w"], a[k]["seedcov"] = int(rng.choice([100, 100, 200])), int(rng.integers(19, qe - qb + 1))
                a[k]["csub"] = int(rng.choice([0, 0, 0, 20, sc - 3 if sc > 3 else 0]))
        out.append(a)
    return out


def paired_vectors(rng, n_pairs, l_pac, L=150):
    """2*n_pairs vectors; mates mostly FR at an insert of 200-500, some improper, some with several hits per end."""
    out = []
    for _ in range(n_pairs):
        pos = int(rng.integers(5000, l_pac - 5000))
        ins = int(rng.integers(200, 500)) if rng.random() < 0.9 else int(rng.integers(2000, 90000))
        ends = []
        for r in range(2):
            n = int(rng.choice([0, 1, 1, 1, 2, 3, 6]))
            a = np.zeros(n, dtype=kswlib.ALNREG)
            for k in range(n):
                jitter = 0 if k == 0 else int(rng.choice([0, 7, 300, 40000]))
                if r == 0:
                    rb = pos + jitter
                else:  # mate on the reverse strand: doubled coordinate
                    fwd_end = pos + ins + jitter
                    rb = 2 * l_pac - fwd_end
                sc = int(rng.choice([150, 150, 140, 120, 60, 25]))
                a[k]["rb"], a[k]["re"], a[k]["qb"], a[k]["qe"] = rb, rb + L, 0, L
                a[k]["score"], a[k]["truesc"], a[k]["w"], a[k]["seedcov"] = sc, sc, 100, int(rng.integers(19, L))
                a[k]["secondary"] = -1
            # the pairing code expects vectors as mem_mark_primary_se leaves them: best score first
            a = a[np.argsort(-a["score"], kind="stable")]
            ends.append(a)
        if rng.random() < 0.1:
            ends.reverse()
        out.extend(ends)
    return out
