#!/usr/bin/env python3
"""tools/bench_se1m.py -- round 1's bench line, kept for continuity (`python bench.py --workload se1m` runs it): one flat
bmh_extend_batch_device() call over the extension tasks of BASELINE.json configs[1] (1 M synthetic SE reads), left and
right extensions in the same launch with a guessed right h0, plus secondary measurements of the other kernels.
The contract line of this repository is bench.py's default workload (configs[2], the whole DP path).

A "step" is one pass of the hot path over one batch of synthetic input: the extension
tasks mem_chain2aln would build for `--reads` simulated reads (per GPU), already resident
in HBM when the timed region starts, run by ONE bmh_extend_batch_device() call through the
C-ABI of libbwamem_hip.so.  Weak scaling: every rank owns its own batch, no collective on
the data path; torch.distributed is used only for the barrier and the
max-over-ranks timing.

Prints ONE JSON line (rank 0).  metric = BASELINE.json's "aligned reads/sec".

Beside the contract fields the line carries, at N=1, three secondary measurements of the other kernels on the path and
around it (each with its own parity check and CPU figure): `global_alignment` (ksw_global2 + traceback),
`mate_rescue_sw` (ksw_align2) and `seeding_fmindex` (bwt_smem1 / bwt_sa; needs oracle/_ref to build an index).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=1_000_000, help="simulated reads per GPU per step")
    ap.add_argument("--workload", "--shape", dest="workload", default="150bp", choices=["150bp", "250bp", "mixed100-300"])
    ap.add_argument("--sw-tasks", type=int, default=400_000,
                    help="mate-rescue Smith-Waterman tasks for the secondary measurement (0 = skip)")
    ap.add_argument("--seed-reads", type=int, default=200_000,
                    help="reads for the secondary FM-index (seeding) measurement; needs oracle/_ref to build an index (0 = skip)")
    ap.add_argument("--target-source", default="pool", choices=["pool", "pac"],
                    help="pac: targets decoded on the fly from a 2-bit reference resident in HBM (BMH_F_TPAC)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (rank 0, N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--global-tasks", type=int, default=250_000,
                    help="size of the secondary ksw_global2 measurement (0 = skip); rank 0 at N=1 only")
    ap.add_argument("--oversubscribe", action="store_true")
    args = ap.parse_args(argv)

    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    import kswlib  # record layouts + the oracle binding (checker / cpu_baseline only)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # under torch.distributed.run (RANK set) the process group is always created -- also for one rank -- so the
    # RCCL path is the same code at N=1,2,4,8; a bare `python bench.py` stays single-process
    use_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if world != args.gpus:
        sys.exit(f"bench_se1m: --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --workload se1m --gpus N` "
                 f"(bench.py launches its own ranks) or under torch.distributed.run")
    if args.oversubscribe:  # rehearsal: more ranks than devices, report over gloo (bench.py --oversubscribe)
        local_rank = local_rank % max(1, torch.cuda.device_count())
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.oversubscribe:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    pkg = load_package()
    import importlib
    tg = importlib.import_module("bwa_mem_quickassist_amd.taskgen")
    sh = importlib.import_module("bwa_mem_quickassist_amd.shard")

    # ---- workload: this rank's shard (distinct seed per rank, same size: weak scaling)
    params = kswlib.make_params()  # bwa mem defaults, reference bwamem.c:45-75
    t0 = time.time()
    pool, tasks, tread = tg.generate(params, args.reads, args.workload, seed=sh.shard_seed(7, rank))
    n_tasks = len(tasks)
    n_reads_used = int(len(np.unique(tread)))
    gen_s = time.time() - t0
    pac, l_pac = None, 0
    if args.target_source == "pac":
        # the byte pool doubles as the forward strand of a synthetic genome: same tasks, same answers, but the
        # kernels fetch target bases from the packed copy (position on the doubled coordinate = pool offset)
        l_pac = len(pool)
        q = np.concatenate([pool & 3, np.zeros((-l_pac) % 4 + 4, np.uint8)])
        q = q[: len(q) // 4 * 4].reshape(-1, 4)
        pac = (q[:, 0] << 6 | q[:, 1] << 4 | q[:, 2] << 2 | q[:, 3]).astype(np.uint8)
        tasks["flags"] |= pkg.BMH_F_TPAC
    alg_bytes = int(tasks["qlen"].astype(np.int64).sum() + tasks["tlen"].astype(np.int64).sum() + 56 * n_tasks)

    d_pool = torch.from_numpy(pool).to(dev)
    d_tasks = torch.from_numpy(tasks.view(np.uint8)).to(dev)
    d_res = torch.zeros(n_tasks * pkg.EXT_RES.itemsize, dtype=torch.uint8, device=dev)

    ctx = pkg.Context(local_rank, params)
    ctx.set_qcap(int(tasks["qlen"].max()))
    if pac is not None:
        ctx.set_pac(pac, l_pac)
    # a dedicated (non-null) torch stream: the kernel is launched on it through the C-ABI and the
    # HIP events that time it are recorded on the same stream
    stream = torch.cuda.Stream(dev)
    assert stream.cuda_stream != 0
    ctx.set_stream(stream.cuda_stream)
    torch.cuda.synchronize(dev)

    def step():
        ctx.extend_batch_device(d_pool.data_ptr(), d_tasks.data_ptr(), n_tasks, d_res.data_ptr())

    def barrier():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()

    ctx.set_kernel_timing(True)  # HIP events around each extension kernel, on the launch stream
    for _ in range(args.warmup):
        step()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_start = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t_start
    step_kernels_ms = ev0.elapsed_time(ev1) / args.steps  # all kernels of a step, HIP events on the launch stream
    bin_ms = ctx.last_extend_bin_ms()                      # per kernel, last timed step
    ctx.sync()  # surfaces any BMH_E_RANGE flagged by the kernel
    if use_dist:
        dist.barrier()
    elapsed, reads_all, tasks_all = sh.reduce_report(elapsed, n_reads_used, n_tasks, torch.device("cpu") if args.oversubscribe else dev)

    # ---- secondary measurement: the banded global alignment + traceback kernel (row a2), N=1 only
    glb = None
    if world == 1 and args.global_tasks > 0:
        gpool, gtasks, gwords = tg.generate_global(args.global_tasks, args.workload, seed=11)
        dg_pool = torch.from_numpy(gpool).to(dev)
        dg_tasks = torch.from_numpy(gtasks.view(np.uint8)).to(dev)
        dg_res = torch.zeros(len(gtasks) * pkg.GLB_RES.itemsize, dtype=torch.uint8, device=dev)
        dg_cig = torch.zeros(gwords + 4, dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)  # fills run on torch's current stream, the kernels on the context's
        ctx.set_qcap(int(max(gtasks["qlen"].max(), tasks["qlen"].max())))
        gsteps = max(3, args.steps // 4)
        with torch.cuda.stream(stream):
            ctx.global_batch_device(dg_pool.data_ptr(), dg_tasks.data_ptr(), len(gtasks), dg_res.data_ptr(), dg_cig.data_ptr())
            torch.cuda.synchronize(dev)
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g0.record(stream)
            for _ in range(gsteps):
                ctx.global_batch_device(dg_pool.data_ptr(), dg_tasks.data_ptr(), len(gtasks), dg_res.data_ptr(),
                                        dg_cig.data_ptr())
            g1.record(stream)
            torch.cuda.synchronize(dev)
        ctx.sync()
        g_ms = g0.elapsed_time(g1) / gsteps
        gres = dg_res.cpu().numpy().view(pkg.GLB_RES)
        gcig = dg_cig.cpu().numpy().view(np.uint32)
        ncores = os.cpu_count() or 1
        ns = min(len(gtasks), 100000)
        t1 = time.perf_counter()
        ores, ocig, ocells = kswlib.orc_global_batch_mt(params, gpool, gtasks[:ns], gwords, nthreads=ncores)
        t1 = time.perf_counter()
        ores, ocig, ocells = kswlib.orc_global_batch_mt(params, gpool, gtasks[:ns], gwords, nthreads=ncores)
        g_cpu_dt = time.perf_counter() - t1
        ok = bool((ores == gres[:ns]).all())
        for k in range(0, ns, 97):
            o, nn = int(gtasks[k]["cigar_off"]), int(ores[k]["n_cigar"])
            ok = ok and bool((ocig[o:o + nn] == gcig[o:o + nn]).all())
        band_cells = float((np.minimum(gtasks["qlen"].astype(np.int64), 2 * gtasks["w"].astype(np.int64) + 1)
                            * gtasks["tlen"].astype(np.int64)).sum())
        g_bytes = float(gtasks["qlen"].astype(np.int64).sum() + gtasks["tlen"].astype(np.int64).sum()
                        + 40 * len(gtasks) + 4 * gres["n_cigar"].astype(np.int64).sum())
        glb = {"kernel": "global_lane_kernel<64|128> (ksw_global2 + traceback, 64 tasks/wave)", "tasks": int(len(gtasks)), "ms": g_ms,
               "tasks_per_s": len(gtasks) / (g_ms * 1e-3), "band_gcups": band_cells / (g_ms * 1e-3) / 1e9,
               "algorithmic_GBps": g_bytes / (g_ms * 1e-3) / 1e9, "mean_w": float(gtasks["w"].mean()),
               "parity": "bit-exact vs oracle (scores, n_cigar, sampled CIGARs)" if ok else "MISMATCH vs oracle",
               "cpu_port_tasks_per_s": ns / g_cpu_dt, "cpu_threads": ncores}
        del dg_pool, dg_tasks, dg_res, dg_cig

    # ---- secondary measurement: mate-rescue local Smith-Waterman (ksw_align2), N=1 only
    swb = None
    if world == 1 and args.sw_tasks > 0:
        spool, stasks = tg.generate_sw(params, args.sw_tasks, args.workload, seed=13)
        ds_pool = torch.from_numpy(spool).to(dev)
        ds_tasks = torch.from_numpy(stasks.view(np.uint8)).to(dev)
        ds_res = torch.zeros(len(stasks) * pkg.SW_RES.itemsize, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)
        ssteps = max(3, args.steps // 4)
        with torch.cuda.stream(stream):
            ctx.sw_batch_device(ds_pool.data_ptr(), ds_tasks.data_ptr(), len(stasks), ds_res.data_ptr())
            torch.cuda.synchronize(dev)
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record(stream)
            for _ in range(ssteps):
                ctx.sw_batch_device(ds_pool.data_ptr(), ds_tasks.data_ptr(), len(stasks), ds_res.data_ptr())
            s1.record(stream)
            torch.cuda.synchronize(dev)
        ctx.sync()
        s_ms = s0.elapsed_time(s1) / ssteps
        sres = ds_res.cpu().numpy().view(pkg.SW_RES)
        ncores = os.cpu_count() or 1
        ns = min(len(stasks), 4000)
        want, _ = kswlib.orc_sw_batch(params, spool, stasks[:ns], nthreads=ncores)
        ok = all(bool((want[f] == sres[:ns][f]).all()) for f in kswlib.SW_FIELDS)
        # cells as the reference visits them: qlen columns x rows of the first pass (all tlen rows unless it stops)
        # + rows of the reversed pass (te - tb + 1); counted from the results, so it is implementation independent
        ql, tl = stasks["qlen"].astype(np.int64), stasks["tlen"].astype(np.int64)
        second = sres["tb"] >= 0
        cells = float((ql * tl).sum() + ((sres["qe"].astype(np.int64) + 1) * (sres["te"] - sres["tb"] + 1))[second].sum())
        sw_word = bool((stasks["xtra"] & 0x10000).sum() * 2 < len(stasks))  # KSW_XBYTE absent: ksw_i16's layout
        sw_cols = int(stasks["qlen"].max())
        swb = {"kernel": "sw_lane_kernel<%d%s> (ksw_align2 %s mode, 64 tasks/wave, packed u16)"
                         % (128 if sw_word or sw_cols > 160 else 80 if sw_cols > 80 else 40, ", WORD" if sw_word else "",
                            "word" if sw_word else "byte"), "tasks": int(len(stasks)),
               "ms": s_ms, "tasks_per_s": len(stasks) / (s_ms * 1e-3), "gcups": cells / (s_ms * 1e-3) / 1e9,
               "mean_qlen": float(ql.mean()), "mean_tlen": float(tl.mean()), "rescued": float(second.mean()),
               "parity": "bit-exact vs oracle (kswr_t, %d sampled tasks)" % ns if ok else "MISMATCH vs oracle"}
        if kswlib.have_ref():  # the reference's own SSE2 ksw_align2, compiled into oracle/_ref by oracle/Makefile
            nr = min(len(stasks), 40000)
            kswlib.ref_sw_batch_mt(params, spool, stasks[:2000], nthreads=ncores)
            t1 = time.perf_counter()
            rres = kswlib.ref_sw_batch_mt(params, spool, stasks[:nr], nthreads=ncores)
            r_dt = time.perf_counter() - t1
            same = all(bool((rres[f] == sres[:nr][f]).all()) for f in kswlib.SW_FIELDS)
            swb.update({"cpu_reference_tasks_per_s": nr / r_dt, "cpu_threads": ncores,
                        "cpu_reference_equal": same, "cpu_sample": "%d tasks, reference ksw_align2 (SSE2) on %d threads" % (nr, ncores)})
        del ds_pool, ds_tasks, ds_res

    # ---- secondary measurement: FM-index queries of the seeding stage, N=1 only.  The index is
    # built by the compiled reference (oracle/_ref/bwa index) over a synthetic genome; skipped where oracle/_ref is absent.
    seedb = None
    if world == 1 and args.seed_reads > 0:
        import reflib
        if reflib.have_ref_bwa() and os.path.exists(os.path.join(kswlib.REF_DIR, "libref_seed_shim.so")):
            import ctypes as C
            import tempfile
            rng = np.random.default_rng(20261010)
            tmpd = tempfile.mkdtemp(prefix="bmh_seedb_")
            G = 8_000_000
            gref = rng.integers(0, 4, G, dtype=np.uint8)
            fa = os.path.join(tmpd, "ref.fa")
            reflib.write_fasta(fa, "synth", gref)
            reflib.build_index(fa)
            idx = reflib.lib().bwa_idx_load(fa.encode(), 7)
            prim, L2, sl, words, sai, sa = reflib.bwt_arrays(idx)
            so = reflib.smem_opt_of(reflib.opt_from_params(params))
            Lr, nr = 150, args.seed_reads
            pos = rng.integers(0, G - Lr - 8, size=nr)
            sreads = gref[pos[:, None] + np.arange(Lr)[None, :]]
            sub = rng.random(sreads.shape) < 0.02
            sreads = np.where(sub, (sreads + rng.integers(1, 4, sreads.shape)) & 3, sreads).astype(np.uint8)
            rcm = rng.random(nr) < 0.5
            sreads[rcm] = 3 - sreads[rcm][:, ::-1]
            rl = list(sreads)
            ctx.set_bwt(prim, L2, sl, words, sai, sa)
            ctx.smem_batch(so, rl[:2000])
            got = ctx.smem_batch(so, rl)
            k_smem = ctx.last_kernel_ms()
            keep = []
            cb = kswlib.make_cbwt(prim, L2, sl, words, sai, sa, keep)
            orc = kswlib.load_oracle()
            orc.orc_fm_extends.restype = C.c_uint64
            orc.orc_fm_extends(1)
            nsmp = min(nr, 2000)
            okf = True
            for r in range(nsmp):
                wc, wi = kswlib.orc_smem_calls(cb, so, rl[r])
                gc, gi = got[r]
                okf = okf and len(gc) == len(wc) and len(gi) == len(wi) and bool((gi == wi).all()) and bool((gc["ret"] == wc["ret"]).all())
            ext_per_read = orc.orc_fm_extends(1) / nsmp
            iv = np.concatenate([x for _, x in got])
            sl_ = (iv["info"] & 0xffffffff).astype(np.int64) - (iv["info"] >> 32).astype(np.int64)
            sel = (sl_ >= int(so["min_seed_len"])) & (iv["x2"] <= 10000)
            x0s, x2s = iv["x0"][sel].astype(np.int64), iv["x2"][sel].astype(np.int64)
            ks = (np.repeat(x0s, x2s) + (np.arange(int(x2s.sum()), dtype=np.int64) - np.repeat(np.cumsum(x2s) - x2s, x2s))).astype(np.uint64)
            ctx.sa_batch(ks[:1000])
            posg = ctx.sa_batch(ks)
            k_sa = ctx.last_kernel_ms()
            shim = C.CDLL(os.path.join(kswlib.REF_DIR, "libref_seed_shim.so"))
            shim.ref_smem_iter_mt.restype = C.c_uint64
            ncores = os.cpu_count() or 1
            spool = np.ascontiguousarray(sreads.reshape(-1))
            off = np.arange(nr, dtype=np.uint64) * Lr
            lens = np.full(nr, Lr, dtype=np.int32)
            cs = C.c_uint64(0)
            bwt_p = C.c_void_p(idx.contents.bwt)
            a_ = (bwt_p, C.c_int(nr), spool.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p),
                  C.c_int(int(so["split_len"])), C.c_int(int(so["split_width"])), C.c_int(int(so["start_width"])), C.c_int(ncores), C.byref(cs))
            shim.ref_smem_iter_mt(*a_)
            t1 = time.perf_counter()
            shim.ref_smem_iter_mt(*a_)
            cpu_smem = time.perf_counter() - t1
            posc = np.zeros(len(ks), dtype=np.uint64)
            shim.ref_sa_mt(bwt_p, ks.ctypes.data_as(C.c_void_p), C.c_int(len(ks)), posc.ctypes.data_as(C.c_void_p), C.c_int(ncores))
            t1 = time.perf_counter()
            shim.ref_sa_mt(bwt_p, ks.ctypes.data_as(C.c_void_p), C.c_int(len(ks)), posc.ctypes.data_as(C.c_void_p), C.c_int(ncores))
            cpu_sa = time.perf_counter() - t1
            seedb = {"kernel": "smem_kernel (bwt_smem1 in smem_next2 order, one lane per read) + sa_kernel (bwt_sa)",
                     "genome_bp": G, "reads": nr, "smem_kernel_ms": k_smem, "reads_per_s": nr / (k_smem * 1e-3),
                     "bwt_extend_per_read": ext_per_read,
                     "algorithmic_GBps": ext_per_read * 2 * 64.0 * nr / (k_smem * 1e-3) / 1e9, "hbm_frac": ext_per_read * 2 * 64.0 * nr / (k_smem * 1e-3) / 8e12,
                     "sa_lookups": int(len(ks)), "sa_kernel_ms": k_sa, "sa_lookups_per_s": len(ks) / (k_sa * 1e-3),
                     "parity": ("bit-exact vs oracle (%d reads) and vs the reference's bwt_sa (all look-ups)" % nsmp)
                     if okf and bool((posc == posg).all()) else "MISMATCH",
                     "cpu_reference_reads_per_s": nr / cpu_smem, "cpu_reference_sa_lookups_per_s": len(ks) / cpu_sa, "cpu_threads": ncores}

    # ---- parity spot-check + CPU baseline (untimed w.r.t. the GPU figure)
    res = d_res.cpu().numpy().view(pkg.EXT_RES)
    out = None
    if rank == 0:
        ncores = os.cpu_count() or 1
        cpu = None
        sample_n = min(n_tasks, 20000)
        want, cells = kswlib.orc_extend_batch(params, pool, tasks[:sample_n], nthreads=ncores, pac=pac, l_pac=l_pac)
        parity_ok = bool((want == res[:sample_n]).all())
        cells_per_task = cells / max(sample_n, 1)
        if world == 1 and not args.no_cpu_baseline:
            rate = None
            t1 = time.perf_counter()
            kswlib.orc_extend_batch(params, pool, tasks[:sample_n], nthreads=ncores, pac=pac, l_pac=l_pac)
            rate = sample_n / (time.perf_counter() - t1)
            big = int(min(n_tasks, max(sample_n, rate * args.cpu_seconds)))
            reps = max(1, int(rate * args.cpu_seconds / big))  # whole passes over the sample, ~cpu_seconds in all
            t1 = time.perf_counter()
            for _ in range(reps):
                want2, cells2 = kswlib.orc_extend_batch(params, pool, tasks[:big], nthreads=ncores, pac=pac, l_pac=l_pac)
            dt = time.perf_counter() - t1
            parity_ok = parity_ok and bool((want2 == res[:big]).all())
            cells_per_task = cells2 / big
            reads_in_sample = len(np.unique(tread[:big]))
            cpu = {"value": reads_in_sample * reps / dt, "unit": "reads/s", "cores": ncores, "kind": "port",
                   "sample": f"{reps} pass(es) over the first {big} extension tasks ({reads_in_sample} reads) of the "
                             f"same batch, oracle/ksw_oracle.c on {ncores} pthreads, {dt:.1f} s",
                   "tasks_per_s": big * reps / dt, "gcups": cells2 * reps / dt / 1e9}
        ms_per_step = elapsed / args.steps * 1e3
        value = reads_all * args.steps / elapsed
        # dominant kernel = the length bin that takes the most time; its algorithmic bytes / its duration
        mode = os.environ.get("BMH_EXT_MODE", "lane")
        fam = {"lane": ["extend_lane_kernel<32> (qlen<=32, 64 tasks/wave)", "extend_lane_kernel<64> (qlen<=64, 64 tasks/wave)",
                        "extend_lane_kernel<128> (qlen<=128, 64 tasks/wave)"],
               "grp": ["extend_grp_kernel<2> (qlen<=32, 4 tasks/wave)", "extend_grp_kernel<4> (qlen<=64, 4 tasks/wave)",
                       "extend_grp_kernel<8> (qlen<=128, 4 tasks/wave)"],
               "reg": ["extend_reg_kernel<1> (qlen<=32)", "extend_reg_kernel<1> (qlen<=64)", "extend_reg_kernel<2> (qlen<=128)"],
               "lds": ["-", "-", "-"]}[mode if mode != "lanex4" else "lane"]
        bin_names = fam + (["extend_lanex_kernel<2> (qlen<=256, 32 tasks/wave)", "extend_lanex_kernel<4> (qlen<=512, 16 tasks/wave)"]
                           if mode in ("lane", "lanex4") else ["extend_reg_kernel<4> (qlen<=256)", "-"]) + ["extend_lds_kernel (longer)"]
        ql = tasks["qlen"].astype(np.int64)
        tl = tasks["tlen"].astype(np.int64)
        which = np.where(ql < 1, 5, np.where(ql <= 32, 0, np.where(ql <= 64, 1, np.where(ql <= 128, 2, np.where(
            ql <= 256, 3, np.where((ql <= 512) & (mode == "lanex4"), 4, 5))))))
        if mode == "grp":
            which = np.where((ql >= 1) & (ql <= 256) & (tl > 1024), 3, which)
        if mode == "lds":
            which[:] = 5
        per_task_bytes = ql + tasks["tlen"].astype(np.int64) + 56  # qlen + tlen + 32 + 24
        kernels = []
        for b in range(6):
            nb = int((which == b).sum())
            if nb == 0:
                continue
            bb = int(per_task_bytes[which == b].sum())
            kernels.append({"kernel": bin_names[b], "tasks": nb, "ms": bin_ms[b], "algorithmic_bytes": bb,
                            "GBps": bb / (bin_ms[b] * 1e-3) / 1e9 if bin_ms[b] > 0 else None})
        dom = max(kernels, key=lambda k: k["ms"])
        ach = dom["GBps"]
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):  # HBM bytes per launch from rocprofv3 PMC passes of this same command
            tj = json.load(open(tpath))
            for k, v in tj.get("kernels", {}).items():
                if dom["kernel"].split(" ")[0].rstrip(">") in k:  # e.g. "extend_lane_kernel<128" in "bmh::extend_lane_kernel<128, true>"
                    traffic = v.get("hbm_bytes_per_launch")
        # `metric` is BASELINE.json's, verbatim; what is timed is named in config.workload (BASELINE.json configs[1]: the
        # hg38 configurations need an index this image cannot build, SURVEY.md §8d replaces them by the task generator)
        metric = "aligned reads/sec (150 bp PE vs hg38) at 1/2/4/8 MI355X; SAM bit-exact vs CPU"
        bpath = os.path.join(ROOT, "BASELINE.json")
        if os.path.exists(bpath):
            metric = json.load(open(bpath)).get("metric", metric)
        out = {
            "metric": metric,
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": f"se1m -- BASELINE.json configs[1] shape: {args.reads} x {args.workload} synthetic SE reads per GPU per step -> "
                                   f"the ksw_extend2 tasks mem_chain2aln builds for them (taskgen.c, SURVEY.md §8d), extension hot path "
                                   f"on the GPU; results checked bit-exact against the oracle after timing (the tasks the CPU baseline replays -- the whole batch "
                                   f"on a 256-thread box -- or 20 000 without it)",
                       "reads_per_gpu": args.reads, "tasks_per_gpu": n_tasks,
                       "mean_qlen": float(tasks["qlen"].mean()), "mean_tlen": float(tasks["tlen"].mean()),
                       "target_source": args.target_source, "parallelism": f"static shard x{world}, no collective"},
            "tasks_per_s": tasks_all * args.steps / elapsed,
            "gcups": cells_per_task * tasks_all * args.steps / elapsed / 1e9,
            "parity": "bit-exact vs oracle on sampled tasks" if parity_ok else "MISMATCH vs oracle",
            "roofline": {"bound": "hbm", "bound_measured": "valu-issue", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": dom["kernel"], "kernel_ms": dom["ms"],
                         "algorithmic_bytes_per_launch": dom["algorithmic_bytes"],
                         "step_kernels_ms": step_kernels_ms, "kernels": kernels,
                         "note": "integer max-plus DP: VALU-issue bound, not HBM bound (~220 int-ops per "
                                 "algorithmic byte vs ~5 ops/B machine balance, SURVEY.md §8d); the HBM fraction "
                                 "is reported because the contract asks for it, GCUPS is the honest figure"},
            "cpu_baseline": cpu,
            "global_alignment": glb,
            "mate_rescue_sw": swb,
            "seeding_fmindex": seedb,
            "setup": {"taskgen_s": gen_s},
        }
        if not parity_ok:
            out["value"] = 0.0  # a fast kernel with different results is not done
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
