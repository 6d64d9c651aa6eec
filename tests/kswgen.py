"""Seeded generators of extension / global-alignment test cases.

Covers the matrix of SURVEY.md §8c: default 150 bp flanks, 100-300 bp with
long indels / chimeric tails / N's, narrow bands that force the 2w retry,
small and disabled z-drop, asymmetric gap costs, scaled scores, h0 edge
values, qlen=1 / tlen=0..2, all-N queries, the beg>0 first-column quirk,
mj / max_ie ties (low-complexity sequence), band re-growth and empty rows
(tlen > qlen + w).
"""
import numpy as np

from kswlib import (EXT_TASK, GLB_TASK, SW_TASK, BMH_F_QREV, BMH_F_TREV, BMH_F_QCOMP, KSW_XBYTE, KSW_XSTOP, KSW_XSUBO,
                    KSW_XSTART, make_params, fill_scmat)


def rand_seq(rng, n, p_n=0.0, alphabet=4):
    s = rng.integers(0, alphabet, size=n, dtype=np.uint8)
    if p_n > 0 and n:
        s[rng.random(n) < p_n] = 4
    return s


def mutate(rng, seq, sub=0.02, ins=0.0025, dele=0.0025, max_indel=1):
    """Copy of `seq` with substitutions and indels (lengths 1..max_indel)."""
    out = []
    i, n = 0, len(seq)
    while i < n:
        r = rng.random()
        if r < sub:
            out.append((int(seq[i]) + int(rng.integers(1, 4))) & 3)
            i += 1
        elif r < sub + ins:
            out.extend(rng.integers(0, 4, size=int(rng.integers(1, max_indel + 1))).tolist())
        elif r < sub + ins + dele:
            i += int(rng.integers(1, max_indel + 1))
        else:
            out.append(int(seq[i]))
            i += 1
    return np.array(out, dtype=np.uint8)


class PoolBuilder:
    """Byte pool + task records; can store a sequence reversed and flag it."""

    def __init__(self, dtype):
        self.chunks, self.size, self.tasks, self.dtype = [], 0, [], dtype

    def put(self, seq, rev=False):
        seq = np.asarray(seq, dtype=np.uint8)
        start = self.size
        self.chunks.append(seq[::-1] if rev else seq)
        self.size += len(seq)
        # forward: offset of base 0; reversed storage: base 0 sits at the END of the stored run
        return (start + len(seq) - 1 if len(seq) else start) if rev else start

    def finish(self):
        pool = np.concatenate(self.chunks + [np.zeros(8, np.uint8)]) if self.chunks else np.zeros(8, np.uint8)
        return pool, np.array(self.tasks, dtype=self.dtype)


def _add_ext(pb, rng, q, t, h0, w, end_bonus, allow_rev=True):
    qrev = allow_rev and rng.random() < 0.3
    trev = allow_rev and rng.random() < 0.3
    qo, to = pb.put(q, qrev), pb.put(t, trev)
    flags = (BMH_F_QREV if qrev else 0) | (BMH_F_TREV if trev else 0)
    pb.tasks.append((qo, to, len(q), len(t), h0, w, end_bonus, flags, 0))


def flank_pair(rng, qlen, gap, sub=0.02, ins=0.0025, dele=0.0025, max_indel=1, p_n=0.0,
               chimera=0.0, alphabet=4):
    """A read flank and the reference window it came from, followed by random bases."""
    q = rand_seq(rng, qlen, p_n, alphabet)
    src = q.copy()
    src[src > 3] = rng.integers(0, 4, size=int((src > 3).sum()), dtype=np.uint8)
    if chimera > 0 and rng.random() < chimera and qlen > 8:
        cut = int(rng.integers(4, qlen))
        src[cut:] = rand_seq(rng, qlen - cut, 0, alphabet)
    t = mutate(rng, src, sub, ins, dele, max_indel)
    t = np.concatenate([t, rand_seq(rng, max(0, qlen + gap - len(t)), 0, alphabet)])[: qlen + gap]
    return q, t


def gen_ext_realistic(rng, n, read_len=(150, 150), hard=False, w=100):
    """Flanks as mem_chain2aln builds them (reference bwamem.c:810-866)."""
    pb = PoolBuilder(EXT_TASK)
    for _ in range(n):
        L = int(rng.integers(read_len[0], read_len[1] + 1))
        slen = int(rng.integers(19, max(20, L - 1)))
        qlen = int(rng.integers(1, L - slen + 1))
        gap = min(2 * w, max(1, qlen - 5)) + int(rng.integers(0, 60))
        if hard:
            q, t = flank_pair(rng, qlen, gap, 0.03, 0.01, 0.01, 12, 0.04, 0.3)
        else:
            q, t = flank_pair(rng, qlen, gap)
        h0 = slen if rng.random() < 0.5 else int(rng.integers(19, L))
        _add_ext(pb, rng, q, t, h0, w, 5)
    return pb.finish()


def gen_ext_fuzz(rng, n, p):
    """Adversarial shapes under parameter set `p`."""
    pb = PoolBuilder(EXT_TASK)
    for _ in range(n):
        kind = int(rng.integers(0, 10))
        alphabet = 4 if kind < 6 else int(rng.integers(1, 3))  # low complexity -> many ties
        qlen = int(rng.integers(1, 200))
        if kind == 0:
            qlen = int(rng.integers(1, 4))
        if kind in (1, 2, 3, 6, 7):  # related sequences
            gap = int(rng.integers(0, 120))
            q, t = flank_pair(rng, qlen, gap, rng.random() * 0.15, rng.random() * 0.05,
                              rng.random() * 0.05, int(rng.integers(1, 20)), rng.random() * 0.1 if kind == 3 else 0,
                              0.3 if kind == 2 else 0, alphabet)
            if kind == 7:  # tlen >> qlen + w: empty rows / j==qlen test
                t = np.concatenate([t, rand_seq(rng, int(rng.integers(50, 250)), 0, alphabet)])
        elif kind == 4:
            q, t = rand_seq(rng, qlen, 1.0), rand_seq(rng, int(rng.integers(0, 60)), 0.2)  # all-N query
        else:
            q, t = rand_seq(rng, qlen, 0.02, alphabet), rand_seq(rng, int(rng.integers(0, 300)), 0.02, alphabet)
        if kind == 0:
            t = t[: int(rng.integers(0, 3))]
        h0 = int(rng.choice([0, 1, int(p["o_ins"]) + int(p["e_ins"]), int(p["o_ins"]) + int(p["e_ins"]) + 1,
                             int(rng.integers(0, 40)), int(rng.integers(19, 200)), int(rng.integers(0, 400))]))
        w = int(rng.choice([1, 2, 5, 10, 20, 50, 100, 200, int(rng.integers(1, 130))]))
        eb = int(rng.choice([0, 5, int(rng.integers(0, 12))]))
        _add_ext(pb, rng, q, t, h0, w, eb)
    return pb.finish()


def fuzz_param_sets(rng, n):
    """Parameter sets: defaults, narrow bands, z-drop variants, asymmetric gaps, -A scaling."""
    sets = [make_params(), make_params(zdrop=20), make_params(zdrop=0), make_params(zdrop=-1),
            make_params(o_del=6, o_ins=4, e_del=1, e_ins=2), make_params(a=2, b=8, o_del=12, o_ins=12, e_del=2, e_ins=2, zdrop=200, pen_clip5=10, pen_clip3=10),
            make_params(o_del=0, o_ins=0), make_params(a=1, b=1, o_del=1, o_ins=1), make_params(w=10), make_params(w=20, zdrop=1000)]
    while len(sets) < n:
        a = int(rng.integers(1, 4))
        p = make_params(a=a, b=int(rng.integers(1, 9)), o_del=int(rng.integers(0, 11)), e_del=int(rng.integers(1, 5)),
                        o_ins=int(rng.integers(0, 11)), e_ins=int(rng.integers(1, 5)), w=int(rng.integers(1, 120)),
                        zdrop=int(rng.choice([-1, 0, 3, 10, 50, 100, 500])),
                        pen_clip5=int(rng.integers(0, 12)), pen_clip3=int(rng.integers(0, 12)))
        if rng.random() < 0.3:  # a fully general (asymmetric) matrix
            p["mat"] = rng.integers(-6, 4, size=25).astype(np.int8)
        sets.append(p)
    return sets[:n]


# ---------------------------------------------------------------- global

def _add_glb(pb, q, t, w, want_cigar=True):
    qo, to = pb.put(q), pb.put(t)
    pb.tasks.append((qo, to, len(q), len(t), w, 0, (len(q) + len(t) + 2) if want_cigar else 0))


def finish_glb(pb):
    pool, tasks = pb.finish()
    off = 0
    for t in tasks:
        t["cigar_off"] = off
        off += int(t["cigar_cap"])
    return pool, tasks, off


def gen_glb_realistic(rng, n, read_len=(150, 150), hard=False):
    """Region pairs as bwa_gen_cigar2 hands them over (reference bwa.c:89-133)."""
    pb = PoolBuilder(GLB_TASK)
    for _ in range(n):
        L = int(rng.integers(read_len[0], read_len[1] + 1))
        qlen = int(rng.integers(max(20, L // 2), L + 1))
        q = rand_seq(rng, qlen, 0.02 if hard else 0.0)
        src = q.copy()
        src[src > 3] = 0
        t = mutate(rng, src, 0.03 if hard else 0.02, 0.01 if hard else 0.0025, 0.01 if hard else 0.0025, 12 if hard else 2)
        if len(t) == 0:
            t = rand_seq(rng, 1)
        w = max(abs(len(q) - len(t)), int(rng.integers(1, 40))) + int(rng.integers(0, 8))
        if rng.random() < 0.1:
            w *= int(rng.integers(2, 5))
        _add_glb(pb, q, t, w, rng.random() < 0.95)
    return finish_glb(pb)


def gen_glb_fuzz(rng, n):
    pb = PoolBuilder(GLB_TASK)
    for _ in range(n):
        kind = int(rng.integers(0, 6))
        alphabet = 4 if kind < 4 else int(rng.integers(1, 3))
        qlen = int(rng.integers(1, 160))
        if kind == 0:
            qlen = int(rng.integers(1, 4))
        if kind in (1, 2, 4):
            q, t = flank_pair(rng, qlen, 0, rng.random() * 0.2, rng.random() * 0.08, rng.random() * 0.08,
                              int(rng.integers(1, 15)), rng.random() * 0.05, 0, alphabet)
            t = t[: max(1, len(t) - int(rng.integers(0, 10)))]
        else:
            q, t = rand_seq(rng, qlen, 0.05, alphabet), rand_seq(rng, int(rng.integers(1, 160)), 0.05, alphabet)
        if kind == 0:
            t = t[: int(rng.integers(1, 4))]
        # the reference's caller always passes w >= |qlen - tlen| (bwa.c:116-125, bwamem.c:884-891);
        # narrower bands make the reference read unwritten backtrack bytes, so we stay in-domain.
        w = abs(len(q) - len(t)) + int(rng.choice([0, 1, 2, 5, 10, 30, 100, 200]))
        _add_glb(pb, q, t, w, rng.random() < 0.9)
    return finish_glb(pb)


# ---- local Smith-Waterman (ksw_align2) ---------------------------------------------------------------------------
def _add_sw(pb, rng, q, t, xtra, allow_flags=True):
    """Stores q/t (optionally reversed and/or complemented in the pool, with the flag that undoes it)."""
    qrev = allow_flags and rng.random() < 0.3
    qcomp = allow_flags and rng.random() < 0.3
    trev = allow_flags and rng.random() < 0.2
    qs = np.where(q < 4, 3 - q, 4).astype(np.uint8) if qcomp else q
    qo, to = pb.put(qs, qrev), pb.put(t, trev)
    flags = (BMH_F_QREV if qrev else 0) | (BMH_F_TREV if trev else 0) | (BMH_F_QCOMP if qcomp else 0)
    pb.tasks.append((qo, to, len(t), len(q), flags, xtra, 0))


def sw_xtra_bwa(p, qlen):
    """xtra exactly as mem_matesw / mem_chain2aln_short build it (reference bwamem_pair.c:147, bwamem.c:529)."""
    a = int(p["a"])
    return KSW_XSUBO | KSW_XSTART | (KSW_XBYTE if qlen * a < 250 else 0) | (19 * a)


def gen_sw_materescue(rng, n, p, read_len=(150, 150), win=(300, 700), hit=0.7, hard=False):
    """Mate-rescue shaped tasks (reference bwamem_pair.c:109-175): the mate against an insert-size window that
    contains a mutated copy of it (or not), sometimes twice (-> score2), sometimes only partly."""
    pb = PoolBuilder(SW_TASK)
    for _ in range(n):
        L = int(rng.integers(read_len[0], read_len[1] + 1))
        W = int(rng.integers(win[0], win[1] + 1)) + L
        t = rand_seq(rng, W)
        if hard and rng.random() < 0.2:
            t = rand_seq(rng, W, alphabet=2)  # low complexity: many ties
        q = rand_seq(rng, L)
        if rng.random() < hit:
            st = int(rng.integers(0, W - L // 2))
            src = t[st: st + L]
            q2 = mutate(rng, src, sub=0.03 if not hard else 0.08, ins=0.004, dele=0.004, max_indel=6 if hard else 2)
            k = int(rng.integers(0, L // 3)) if rng.random() < 0.3 else 0  # unrelated head
            q = np.concatenate([rand_seq(rng, k), q2])[:L]
            if len(q) < L:
                q = np.concatenate([q, rand_seq(rng, L - len(q))])
            if rng.random() < 0.25:  # a second, weaker copy elsewhere in the window
                st2 = int(rng.integers(0, W - L))
                cp = mutate(rng, src, sub=0.1)
                t[st2: st2 + len(cp)] = cp[: W - st2]
        if hard and rng.random() < 0.3:
            q[rng.random(len(q)) < 0.04] = 4
        if hard and rng.random() < 0.1:
            t[rng.random(len(t)) < 0.02] = 4
        _add_sw(pb, rng, q, t, sw_xtra_bwa(p, len(q)))
    return pb.finish()


def gen_sw_fuzz(rng, n, p):
    """Function-level fuzz of ksw_align2: every xtra combination, tiny and ragged sizes, byte/word mode."""
    pb = PoolBuilder(SW_TASK)
    a, mx = int(p["a"]), int(np.max(p["mat"]))
    shift = max(0, -int(np.min(p["mat"])))
    for _ in range(n):
        qlen = int(rng.choice([rng.integers(1, 20), rng.integers(20, 160), rng.integers(100, 300)]))
        tlen = int(rng.choice([rng.integers(1, 30), rng.integers(30, 400), rng.integers(200, 900)]))
        t = rand_seq(rng, tlen)
        if rng.random() < 0.7 and tlen > 5:
            st = int(rng.integers(0, tlen))
            core = mutate(rng, t[st: st + min(qlen, tlen - st)], sub=float(rng.choice([0.0, 0.02, 0.1, 0.3])),
                          ins=0.01, dele=0.01, max_indel=7)
            q = np.concatenate([rand_seq(rng, int(rng.integers(0, max(1, qlen - len(core) + 1)))), core])[:qlen]
            if len(q) < qlen:
                q = np.concatenate([q, rand_seq(rng, qlen - len(q))])
            if rng.random() < 0.2:
                k = int(rng.integers(0, tlen))
                t[k: k + len(core)] = core[: tlen - k]
        else:
            q = rand_seq(rng, qlen)
        if rng.random() < 0.15:
            q[rng.random(qlen) < 0.05] = 4
        if rng.random() < 0.1:
            t[rng.random(tlen) < 0.03] = 4
        thr = int(rng.choice([0, 10, 19, 19 * a, 30, 60]))
        xm = rng.random()
        if xm < 0.6:
            xtra = KSW_XSUBO | KSW_XSTART | thr
        elif xm < 0.7:
            xtra = KSW_XSTART
        elif xm < 0.8:
            xtra = KSW_XSUBO | thr
        elif xm < 0.9:
            xtra = KSW_XSTOP | thr
        else:
            xtra = 0
        if rng.random() < 0.6 and (qlen * mx + shift < 255 or not (xtra & KSW_XSTART)):
            xtra |= KSW_XBYTE  # byte mode; overflow (score 255) only where the reference defines the outcome
        _add_sw(pb, rng, q, t, xtra)
    return pb.finish()


def sw_param_sets(rng, n):
    """Parameter sets for ksw_align2 (o_ins >= 1: the closed-form domain of the GPU kernels)."""
    out = [make_params(), make_params(a=2, b=5), make_params(o_del=4, e_del=2, o_ins=6, e_ins=1),
           make_params(o_del=0, e_del=2, o_ins=1, e_ins=3), make_params(a=3, b=1, o_del=2, o_ins=3, e_ins=2)]
    while len(out) < n:
        m = rng.integers(-6, 7, 25).astype(np.int8)
        m[0], m[6] = 2, 1
        out.append(make_params(a=int(rng.integers(1, 4)), o_del=int(rng.integers(0, 8)), e_del=int(rng.integers(1, 4)),
                               o_ins=int(rng.integers(1, 8)), e_ins=int(rng.integers(1, 4)), mat=m))
    return out[:n]
