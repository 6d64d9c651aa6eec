#!/usr/bin/env python3
"""bench.py -- throughput of BWA-MEM's dynamic-programming hot path on MI355X, on BASELINE.json's metric configuration.

Default workload `pe10m` = BASELINE.json configs[2]: 10 M x 150 bp read pairs (20 M reads) per GPU per step.  The GRCh38
index cannot be built here (no genome data, `bwa index` of 3.1 Gb takes hours), so -- as SURVEY.md §8d / BASELINE.md §3
prescribe -- the kernels are driven by a task generator that reproduces the measured task distribution of the
reference pipeline; per-read DP work does not depend on the genome.  A STEP is one pass of the whole DP path over the
20 M reads, streamed through HBM in chunks, inputs resident when the timed region starts:

  1. seed extension  one fused record per seeded read (bmh_seedext_batch_device, reference bwamem.c:808-866 over
                     ksw_extend2): left extension -> band retry -> clip decision -> RIGHT extension started from the
                     left score the device just computed -> retry -> finished region;   ~1.47 ksw_extend2 per read
  2. global          ksw_global2 + traceback (bmh_global_batch_device, bwa.c:132): 0.85 tasks per read at the measured
                     band distribution (mean w 19.6)
  3. mate rescue     ksw_align2 (bmh_sw_batch_device, bwamem_pair.c:148): --rescue-rate x pairs tasks (measured 2-11 %
                     of the pairs; 4.5 % on the synthetic PE set of profiles/r01_pipeline_pe_*)

`value` = reads of the step / time of the step (a pair counts as two reads), max over ranks.  What is NOT in a step: FM-index
seeding, chaining, SAM text (host stages either side of the path); their cost shows in `cpu_baseline_pipeline`, which
times the reference's own `bwa mem` (all host cores) and the same binary with this library preloaded, SAM-identical.

`--workload se1m` runs round 1's line (tools/bench_se1m.py: one flat ksw_extend2 batch over configs[1]-shaped tasks).
Prints ONE JSON line (rank 0).
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

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
_T0 = time.time()


def note(msg):
    """progress on stderr (the JSON line on stdout stays alone)"""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.time() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


# ---------------------------------------------------------------------------------------------------------------------
# whole-pipeline baseline: the reference's own `bwa mem` vs the same binary with the library preloaded
def _fastq_fixed(path, reads, prefix):
    """FASTQ with fixed-width names: every record has the same length, so the file is one numpy reshape."""
    n, L = reads.shape
    name = np.frombuffer(("".join("@%s%08d\n" % (prefix, i) for i in range(n))).encode(), dtype=np.uint8).reshape(n, -1)
    seq = np.frombuffer(b"ACGTN", dtype=np.uint8)[reads]
    nl = np.full((n, 1), 10, np.uint8)
    plus = np.tile(np.frombuffer(b"+\n", dtype=np.uint8), (n, 1))
    qual = np.full((n, L), ord("I"), np.uint8)
    np.concatenate([name, seq, nl, plus, qual, nl], axis=1).tofile(path)


def pipeline_baseline(n_reads, genome_bp, threads, batch=8192, noisy=0.3, dut=True, index_log=None):
    """REF and DUT reads/s from the reference's own `[M::mem_process_seqs] Processed N reads in ... real sec` lines
    (bwamem.c:1320-1321; index loading excluded), paired-end, SAM compared (minus @PG)."""
    import kswgen
    import reflib
    from __graft_entry__ import load_package
    if not reflib.have_ref_bwa():
        return {"skipped": "oracle/_ref/bwa absent (it is built where /root/reference exists and travels with the tree)"}
    rng = np.random.default_rng(20261101)
    tmp = tempfile.mkdtemp(prefix="bmh_pipe_")
    ref = kswgen.rand_seq(rng, genome_bp)
    fa = os.path.join(tmp, "ref.fa")
    reflib.write_fasta(fa, "synth", ref)
    t0 = time.time()
    reflib.build_index(fa, index_log)
    t_index = time.time() - t0
    n_pairs, L = n_reads // 2, 150
    ins = rng.integers(250, 450, size=n_pairs)
    pos = rng.integers(0, len(ref) - 520, size=n_pairs)
    idx = np.arange(L)[None, :]
    a = ref[pos[:, None] + idx]
    b = ref[(pos + ins - L)[:, None] + idx]
    rate_b = np.where(rng.random(n_pairs) < noisy, 0.12, 0.02)[:, None]
    for arr, rate in ((a, 0.02), (b, rate_b)):
        sub = rng.random(arr.shape) < rate
        arr[sub] = (arr[sub] + rng.integers(1, 4, size=int(sub.sum()))) & 3
    b = 3 - b[:, ::-1]
    fq = [os.path.join(tmp, "r1.fq"), os.path.join(tmp, "r2.fq")]
    _fastq_fixed(fq[0], a, "p")
    _fastq_fixed(fq[1], b, "p")

    def run(preload, out, t):
        note(f"pipeline baseline: bwa mem -t {t} {'with the library preloaded' if preload else '(reference)'}")
        env = dict(os.environ)
        if preload:
            env["LD_PRELOAD"] = load_package().DROPIN_PATH
            env["BMH_KSW_DROPIN"] = "1"  # every remaining per-call ksw_* goes to the GPU too: all DP on the device
            env["BMH_VERBOSE"] = "1"
        t0 = time.time()
        with open(out, "w") as f:
            try:
                p = subprocess.run([reflib.REF_BWA, "mem", "-t", str(t), "-b", str(batch), fa] + fq, stdout=f, stderr=subprocess.PIPE,
                                   env=env, timeout=300)
            except subprocess.TimeoutExpired as e:
                return {"error": "timed out after 300 s: " + (e.stderr or b"").decode(errors="replace")[-400:]}
        err = p.stderr.decode(errors="replace")
        if p.returncode != 0:
            return {"error": err[-400:]}
        reads = real = cpu_s = 0
        per_chunk = []
        for m in re.finditer(r"Processed (\d+) reads in ([\d.]+) CPU sec, ([\d.]+) real sec", err):
            reads += int(m.group(1))
            cpu_s += float(m.group(2))
            real += float(m.group(3))
            per_chunk.append((int(m.group(1)), float(m.group(3))))
        later = per_chunk[1:]  # the first chunk of a run also pays for contexts, workspaces and heaps growing to size
        chunks = [l.split("] ", 1)[1] for l in err.splitlines() if l.startswith("[bwamem_hip] chunk of")]
        tail = [l.split("] ", 1)[1] for l in err.splitlines() if l.startswith(("[bwamem_hip] phase", "[bwamem_hip] seeding batch", "[bwamem_hip] mate rescue"))]
        drv = [l.split("] ", 1)[1] for l in err.splitlines() if l.startswith(("[bwamem_hip] bmh_sam_batch", "[bwamem_hip] bmh_reg2cigar_batch", "[bwamem_hip] bmh_chain2aln_batch"))]
        return {"reads": reads, "real_s": real, "reads_per_s": reads / real if real else None, "wall_s": time.time() - t0, "cpu_s": cpu_s,
                "chunk_real_s": [c[1] for c in per_chunk],
                "reads_per_s_after_first_chunk": (sum(c[0] for c in later) / sum(c[1] for c in later)) if later and sum(c[1] for c in later) > 0 else None,
                "chunks": chunks[:8] if preload else None, "thread_seconds": tail[-5:] if preload else None, "driver_trace": drv[-48:-40] + drv[-6:] if drv else None}

    if dut:
        run(True, os.path.join(tmp, "warm.sam"), min(threads, 16))  # page the HIP runtime + code objects in (seconds on a fresh box)
    r = run(False, os.path.join(tmp, "ref.sam"), threads)
    d = run(True, os.path.join(tmp, "dut.sam"), threads) if dut else {"error": "not run"}
    same = None
    if "error" not in r and "error" not in d:
        same = [l for l in open(os.path.join(tmp, "ref.sam")) if not l.startswith("@PG")] == \
               [l for l in open(os.path.join(tmp, "dut.sam")) if not l.startswith("@PG")]
    out = {"value": r.get("reads_per_s"), "unit": "reads/s", "cores": threads, "kind": "reference",
           "sample": f"{n_reads} x 150 bp paired-end reads vs a {genome_bp} bp synthetic genome, oracle/_ref/bwa mem -t {threads} -b {batch}; "
                     f"reads/s from the reference's own per-chunk 'Processed ... real sec' lines",
           "dut_value": d.get("reads_per_s"), "dut": "same binary, LD_PRELOAD=libbwamem_hip_dropin.so, BMH_KSW_DROPIN=1, same -t",
           "dut_over_ref": (d["reads_per_s"] / r["reads_per_s"]) if r.get("reads_per_s") and d.get("reads_per_s") else None,
           "dut_value_after_first_chunk": d.get("reads_per_s_after_first_chunk"), "ref_cpu_s": r.get("cpu_s"), "dut_cpu_s": d.get("cpu_s"),
           "sam_identical": same, "ref": r, "dut_detail": d, "index_s": t_index}
    for f in os.listdir(tmp):
        os.unlink(os.path.join(tmp, f))
    os.rmdir(tmp)
    return out


def host_cores():
    """(host cores this process can actually use, what limits them): the affinity mask cut to the cgroup's CPU quota --
    on the 1-GPU boxes of this pool 256 hardware threads are visible but the container is entitled to 16 CPUs' worth,
    and more runnable threads than that only get throttled"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    note = f"{n} hardware threads in the affinity mask"
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = int(q) / int(per)
            note += f", cgroup cpu.max {q}/{per} = {quota:.1f} CPUs"
            n = max(1, min(n, int(quota + 0.999)))
    except Exception:
        pass
    return n, note + f" -> {n} threads used"


# ---------------------------------------------------------------------------------------------------------------------
def launch_ranks(n, argv):
    """`python bench.py --gpus N` with N > 1 and no RANK in the environment: start N fresh rank processes (one per GPU; static
    contiguous shard per rank like kt_for_batch's ranges, reference kthread_batch.c:44-56, bwamem.c:1313) BEFORE this process has
    made any GPU call, relay rank 0's JSON line, and exit with the worst exit code.  Equivalent to
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`,
    which is what the driver itself uses and which takes the other branch (RANK is set)."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode]
    for p in procs[1:]:
        try:
            rcs.append(p.wait(timeout=120 if rcs[0] == 0 else 5))
        except subprocess.TimeoutExpired:
            p.kill()  # exactly the process started above
            rcs.append(p.wait())
    for line in out.decode().splitlines():  # rank 0's JSON line alone on stdout; anything a library printed there (gloo does) to stderr
        print(line, file=sys.stdout if line.startswith("{") else sys.stderr)
    sys.stdout.flush()
    bad = [rc for rc in rcs if rc != 0]
    if bad:
        print(f"[bench] rank exit codes {rcs}", file=sys.stderr)
    return bad[0] if bad else 0


class HostFeed:
    """--feed host: the step with the PCIe feed inside it.  What travels is what the library's drivers ship (host/chain2aln_batch.c,
    host/reg2cigar_batch.c through the preload shim): per chunk the READS and the task records up, the finished regions, global scores,
    CIGAR words (16 words of room per task, as host/sam_post.c hands it out) and rescue results down.  The reference windows of the
    seed records do not travel: they stand for reference bases, which the shim keeps resident in HBM (bmh_ctx_set_pac; here: the
    windows region of each chunk's pool, uploaded once before the timed region).  Nor do the global tasks' TARGET bytes: they are
    reference windows too, which the library's region record (bmh_region_cigar_batch) fetches from the resident
    reference on the device -- of a global task its query bytes and its 32-byte record travel (--feed-global-windows ships the targets as
    well, as the CIGAR driver did before the region record).  Everything starts in PINNED host memory; each chunk's upload rides a copy
    stream one chunk ahead of the compute streams, the kernels run exactly as in the resident step, results come back on a third
    stream -- all inside the timed region.  Chunks are numbered globally, so the pipeline keeps running across step boundaries."""
    CIGAR_WORDS = 16

    def __init__(self, torch, dev, pkg, tg, host, host_sw, ctxs_of, streams_of, ship_global_windows=False):
        self.torch, self.dev = torch, dev
        self.cx_exts, self.cx_glb, self.cx_sw = ctxs_of
        self.s_exts, self.s_glb, self.s_sw = streams_of
        self.h2d, self.d2h = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        pin = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).pin_memory()
        dz = lambda nbytes: torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        self.chunks = []
        for pool, seeds, gpool, gtasks, gwords in host:
            sp, st, rb = tg.split_reads_windows(pool, seeds)  # [all reads | all windows], offsets rewritten
            d_pool = dz(len(sp))
            d_pool[rb:].copy_(torch.from_numpy(sp[rb:]))  # the windows = the resident reference: uploaded once, outside the timed region
            gp, gt, qb = tg.split_queries_targets(gpool, gtasks)  # [all queries | all targets], offsets rewritten
            if ship_global_windows:
                qb = len(gp)
            d_gpool = dz(len(gp))
            d_gpool[qb:].copy_(torch.from_numpy(gp[qb:]))  # the targets = reference windows: resident, like the seeds' windows
            gt["cigar_off"] = np.arange(len(gt), dtype=np.uint32) * HostFeed.CIGAR_WORDS
            gt["cigar_cap"] = HostFeed.CIGAR_WORDS
            gw = len(gt) * HostFeed.CIGAR_WORDS
            self.chunks.append({"reads": pin(sp[:rb]), "d_pool": d_pool, "gq": pin(gp[:qb]), "d_gpool": d_gpool, "in": [pin(st), pin(gt)],
                                "n": len(st), "ng": len(gt), "window_bytes": len(sp) - rb + len(gp) - qb,
                                "out": [torch.empty(len(st) * pkg.SEED_RES.itemsize, dtype=torch.uint8).pin_memory(),
                                        torch.empty(len(gt) * pkg.GLB_RES.itemsize, dtype=torch.uint8).pin_memory(),
                                        torch.empty((gw + 8) * 4, dtype=torch.uint8).pin_memory()]})
        self.sw = [{"in": [pin(sp_), pin(st_)], "n": len(st_), "out": torch.empty(len(st_) * pkg.SW_RES.itemsize, dtype=torch.uint8).pin_memory()}
                   for sp_, st_ in host_sw]
        cap_in = [max(c["in"][i].numel() for c in self.chunks) for i in range(2)]
        cap_out = [max(c["out"][i].numel() for c in self.chunks) for i in range(3)]
        self.slots = [{"in": [dz(b) for b in cap_in], "out": [dz(b) for b in cap_out]} for _ in range(2)]
        self.sw_slot = {"in": [dz(max(b["in"][i].numel() for b in self.sw)) for i in range(2)], "out": dz(max(b["out"].numel() for b in self.sw))}
        self.h2d_bytes = sum(c["reads"].numel() + c["gq"].numel() + sum(t.numel() for t in c["in"]) for c in self.chunks) + sum(t.numel() for b in self.sw for t in b["in"])
        self.d2h_bytes = sum(t.numel() for c in self.chunks for t in c["out"]) + sum(b["out"].numel() for b in self.sw)
        self.resident_window_bytes = sum(c["window_bytes"] for c in self.chunks)
        self.g = 0            # global chunk number
        self.comp_done = {}   # g -> (event on the extension stream, event on the global stream)
        self.back_done = {}   # g -> event on the d2h stream
        self.up_done = {}
        self.sw_free = None   # the rescue slot: free again once its results are back
        self.copy_events = []  # (start, end, bytes) of uploads when instrumented
        torch.cuda.synchronize(dev)

    def _upload(self, g, timed=False):
        torch = self.torch
        nc = len(self.chunks)
        c, slot = self.chunks[g % nc], self.slots[g % 2]
        for k in {g - 2, g - nc}:  # the slot's previous chunk, and this chunk's own pool one step ago, have been computed
            for ev in self.comp_done.get(k, ()):
                self.h2d.wait_event(ev)
        for k in [k for k in self.comp_done if k < g - max(2, nc)]:
            del self.comp_done[k]
        with torch.cuda.stream(self.h2d):
            if timed:
                e0 = torch.cuda.Event(enable_timing=True); e0.record(self.h2d)
            c["d_pool"][: c["reads"].numel()].copy_(c["reads"], non_blocking=True)
            c["d_gpool"][: c["gq"].numel()].copy_(c["gq"], non_blocking=True)
            for src, dst in zip(c["in"], slot["in"]):
                dst[: src.numel()].copy_(src, non_blocking=True)
            if timed:
                e1 = torch.cuda.Event(enable_timing=True); e1.record(self.h2d)
                self.copy_events.append((e0, e1, c["reads"].numel() + c["gq"].numel() + sum(t.numel() for t in c["in"])))
        ev = torch.cuda.Event(); ev.record(self.h2d)
        self.up_done[g] = ev

    def chunk(self, timed=False):
        """one chunk through the pipeline: upload of the NEXT chunk, compute of this one, download of its results"""
        torch = self.torch
        g = self.g
        if g not in self.up_done:
            self._upload(g, timed)
        c, slot = self.chunks[g % len(self.chunks)], self.slots[g % 2]
        k = g % len(self.chunks)
        s_ext, cx_ext = self.s_exts[k % len(self.s_exts)], self.cx_exts[k % len(self.cx_exts)]
        up = self.up_done.pop(g)
        back = self.back_done.pop(g - 2, None)          # ... and its results have left the slot's result buffers
        for s in (s_ext, self.s_glb):
            s.wait_event(up)
            if back is not None:
                s.wait_event(back)
        cx_ext.seedext_batch_device(c["d_pool"].data_ptr(), slot["in"][0].data_ptr(), c["n"], slot["out"][0].data_ptr())
        self.cx_glb.global_batch_device(c["d_gpool"].data_ptr(), slot["in"][1].data_ptr(), c["ng"], slot["out"][1].data_ptr(), slot["out"][2].data_ptr())
        e_ext, e_glb = torch.cuda.Event(), torch.cuda.Event()
        e_ext.record(s_ext); e_glb.record(self.s_glb)
        self.comp_done[g] = (e_ext, e_glb)
        self._upload(g + 1, timed)  # one chunk ahead of compute
        self.d2h.wait_event(e_ext); self.d2h.wait_event(e_glb)
        with torch.cuda.stream(self.d2h):
            for src, dst in zip(slot["out"], c["out"]):
                dst.copy_(src[: dst.numel()], non_blocking=True)
        eb = torch.cuda.Event(); eb.record(self.d2h)
        self.back_done[g] = eb
        self.g = g + 1

    def rescue(self, b):
        torch = self.torch
        if self.sw_free is not None:
            self.h2d.wait_event(self.sw_free)
        with torch.cuda.stream(self.h2d):
            for src, dst in zip(b["in"], self.sw_slot["in"]):
                dst[: src.numel()].copy_(src, non_blocking=True)
        ev = torch.cuda.Event(); ev.record(self.h2d)
        self.s_sw.wait_event(ev)
        self.cx_sw.sw_batch_device(self.sw_slot["in"][0].data_ptr(), self.sw_slot["in"][1].data_ptr(), b["n"], self.sw_slot["out"].data_ptr())
        e = torch.cuda.Event(); e.record(self.s_sw)
        self.d2h.wait_event(e)
        with torch.cuda.stream(self.d2h):
            b["out"].copy_(self.sw_slot["out"][: b["out"].numel()], non_blocking=True)
        self.sw_free = torch.cuda.Event(); self.sw_free.record(self.d2h)

    def step(self, timed=False):
        for _ in self.chunks:
            self.chunk(timed)
        for b in self.sw:
            self.rescue(b)

    def drain(self):
        self.h2d.synchronize(); self.d2h.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="pe10m", choices=["pe10m", "se1m"])
    ap.add_argument("--shape", default="150bp", choices=["150bp", "250bp", "mixed100-300"], help="read model of the task generator")
    ap.add_argument("--pairs", type=int, default=10_000_000, help="read pairs per GPU per step (pe10m)")
    ap.add_argument("--chunk-reads", type=int, default=10_000_000,
                    help="reads per streamed chunk (round 3: 10 x 2 M 178.1 ms per step, 5 x 4 M 178.2, 2 x 10 M 172.6, 1 x 20 M 178.3 -- more waves per "
                         "launch, fewer half-empty tails; round 2: 20 x 1 M 215, 5 x 4 M 203)")
    ap.add_argument("--global-per-read", type=float, default=0.85, help="ksw_global2 tasks per read (measured, SURVEY.md §8a2)")
    ap.add_argument("--rescue-rate", type=float, default=0.06, help="ksw_align2 mate-rescue tasks per pair (measured 0.02-0.11)")
    ap.add_argument("--rescue-batch", type=int, default=1_000_000, help="mate-rescue tasks are collected over chunks into batches of up to this many")
    ap.add_argument("--ext-contexts", type=int, default=1, help="with --streams 3: alternate the chunks' extension stage over this many contexts/streams")
    ap.add_argument("--streams", type=int, default=3, choices=[1, 3],
                    help="3: one context and HIP stream per stage, so chunk k's global alignments run beside chunk k+1's extensions "
                         "(measured +10 %% reads/s); 1: the stages back to back on one stream (every kernel alone on the chip)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the oracle-port CPU baseline (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline-baseline", action="store_true")
    ap.add_argument("--pipeline-reads", type=int, default=6_400_000, help="reads of the whole-pipeline baseline (six chunks of bwa's at 16 threads)")
    ap.add_argument("--pipeline-batch", type=int, default=32768, help="the fork's -b: reads per phase-1 batch")
    ap.add_argument("--pipeline-genome", type=int, default=4_600_000)
    ap.add_argument("--feed", default="both", choices=["resident", "host", "both"],
                    help="resident: inputs in HBM when the timed region starts (`value`, the contract's definition); host: also time the same steps "
                         "with every chunk uploaded from pinned host memory one chunk ahead of compute and every result downloaded "
                         "(`value_streamed`); both (default): the two timed regions one after the other")
    ap.add_argument("--feed-global-windows", action="store_true",
                    help="host-fed steps: ship the global tasks' target windows as well (the CIGAR driver before the region record); default: they stay "
                         "resident like the seeds' windows")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="rehearsal on a box with fewer GPUs than ranks: rank r uses device r %% (visible devices), the barrier and the report "
                         "go over gloo (RCCL refuses two ranks on one device).  Never for a reported number.")
    ap.add_argument("--rank-echo", action="store_true", help=argparse.SUPPRESS)  # launcher self-test: print this rank's environment, touch no GPU
    args, rest = ap.parse_known_args()
    if args.rank_echo and "RANK" in os.environ:
        if os.environ["RANK"] == "0":
            print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}))
        return 0
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(args.gpus, sys.argv[1:])  # before anything here has touched a GPU
    if args.workload == "se1m":
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_se1m
        return bench_se1m.main(["--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup), "--shape", args.shape] +
                               (["--no-cpu-baseline"] if args.no_cpu_baseline else []) + (["--oversubscribe"] if args.oversubscribe else []) + rest)
    if rest:
        ap.error("unknown arguments: " + " ".join(rest))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    ncores, cores_note = host_cores()

    # ---- whole-pipeline CPU baseline FIRST: it starts child processes, and this process has not touched the GPU yet
    pipe = None
    if world == 1 and rank == 0 and not args.no_pipeline_baseline and args.pipeline_reads > 0:
        try:
            pipe = pipeline_baseline(args.pipeline_reads, args.pipeline_genome, ncores, batch=args.pipeline_batch)
            pipe["cores_note"] = cores_note
        except Exception as e:  # a baseline must never take the measurement down with it
            pipe = {"skipped": f"{type(e).__name__}: {e}"}

    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    import kswlib  # record layouts + the oracle binding (checker / cpu_baseline only)
    import importlib
    from concurrent.futures import ThreadPoolExecutor

    use_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (start it as `python bench.py --gpus N`, which launches its own ranks, "
                 f"or under torch.distributed.run with --nproc-per-node N)")
    dev_index = local_rank
    if args.oversubscribe:
        dev_index = local_rank % max(1, torch.cuda.device_count())  # device_count() does not initialise the GPU on this image
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.oversubscribe:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    rdev = torch.device("cpu") if args.oversubscribe else dev  # where the report's reductions live
    pkg = load_package()
    tg = importlib.import_module("bwa_mem_quickassist_amd.taskgen")
    sh = importlib.import_module("bwa_mem_quickassist_amd.shard")
    params = kswlib.make_params()  # bwa mem defaults, reference bwamem.c:45-75

    # ---- this rank's shard (weak scaling: same size, distinct seeds), generated chunk by chunk on host threads
    n_reads_step = 2 * args.pairs
    n_chunks = max(1, (n_reads_step + args.chunk_reads - 1) // args.chunk_reads)
    sizes = [n_reads_step * (k + 1) // n_chunks - n_reads_step * k // n_chunks for k in range(n_chunks)]
    t0 = time.time()

    def gen(k):
        s0 = sh.shard_seed(7, rank) * 131 + 17 * k
        nr = sizes[k]
        pool, seeds = tg.generate_seeds(params, nr, args.shape, seed=s0)
        gpool, gtasks, gwords = tg.generate_global(max(1, int(round(nr * args.global_per_read))), args.shape, seed=s0 + 5)
        return pool, seeds, gpool, gtasks, gwords

    n_sw_step = max(1, int(round(args.pairs * args.rescue_rate)))
    n_swb = max(1, (n_sw_step + args.rescue_batch - 1) // args.rescue_batch)
    sw_sizes = [n_sw_step * (k + 1) // n_swb - n_sw_step * k // n_swb for k in range(n_swb)]

    def gen_sw(k):
        return tg.generate_sw(params, sw_sizes[k], args.shape, seed=sh.shard_seed(7, rank) * 131 + 9 + 1000 * k)

    workers = max(1, min(16, ncores, n_chunks))
    note(f"generating {n_reads_step} reads' worth of tasks in {n_chunks} chunks on {workers} threads")
    with ThreadPoolExecutor(max_workers=workers) as ex:
        host = list(ex.map(gen, range(n_chunks)))
        host_sw = list(ex.map(gen_sw, range(n_swb)))
    gen_s = time.time() - t0

    note(f"generated in {gen_s:.1f} s; uploading")
    t0 = time.time()
    chunks = []
    up = lambda a: torch.from_numpy(a.view(np.uint8).reshape(-1)).to(dev)
    for pool, seeds, gpool, gtasks, gwords in host:
        chunks.append({"pool": up(pool), "seeds": up(seeds), "n": len(seeds), "gpool": up(gpool), "gtasks": up(gtasks), "ng": len(gtasks),
                       "gwords": gwords,
                       "sres": torch.zeros(len(seeds) * pkg.SEED_RES.itemsize, dtype=torch.uint8, device=dev),
                       "gres": torch.zeros(len(gtasks) * pkg.GLB_RES.itemsize, dtype=torch.uint8, device=dev)})
    swb = [{"spool": up(spool), "stasks": up(stasks), "ns": len(stasks),
            "swres": torch.zeros(len(stasks) * pkg.SW_RES.itemsize, dtype=torch.uint8, device=dev)} for spool, stasks in host_sw]
    d_cig = torch.zeros(max(c["gwords"] for c in chunks) + 8, dtype=torch.int32, device=dev)  # one chunk's CIGARs at a time
    upload_s = time.time() - t0
    n_seeded = sum(c["n"] for c in chunks)
    n_glb, n_sw = sum(c["ng"] for c in chunks), sum(b["ns"] for b in swb)
    qmax = int(max(max(int(h[1]["qbeg"].max()), int((h[1]["l_query"] - h[1]["qbeg"] - h[1]["len"]).max())) for h in host))
    gqmax = int(max(int(h[3]["qlen"].max()) for h in host))

    stream = torch.cuda.Stream(dev)
    assert stream.cuda_stream != 0
    n_ext = max(1, args.ext_contexts) if args.streams == 3 else 1
    ctxs = [pkg.Context(dev_index, params) for _ in range((2 + n_ext) if args.streams == 3 else 1)]
    streams = [stream] + [torch.cuda.Stream(dev) for _ in ctxs[1:]]
    for c, s in zip(ctxs, streams):
        c.set_qcap(max(qmax, gqmax))
        c.set_stream(s.cuda_stream)
    cx_exts, cx_glb, cx_sw = ([ctxs[0]] + ctxs[3:]) if args.streams == 3 else [ctxs[0]], ctxs[1 % len(ctxs)], ctxs[2 % len(ctxs)]
    ext_streams = ([streams[0]] + streams[3:]) if args.streams == 3 else [streams[0]]
    for k, c in enumerate(chunks):
        c["k"] = k
    cx_ext = cx_exts[0]
    torch.cuda.synchronize(dev)

    def run_chunk(c, cig):
        cx_exts[c["k"] % len(cx_exts)].seedext_batch_device(c["pool"].data_ptr(), c["seeds"].data_ptr(), c["n"], c["sres"].data_ptr())
        cx_glb.global_batch_device(c["gpool"].data_ptr(), c["gtasks"].data_ptr(), c["ng"], c["gres"].data_ptr(), cig.data_ptr())

    def run_sw(b):
        cx_sw.sw_batch_device(b["spool"].data_ptr(), b["stasks"].data_ptr(), b["ns"], b["swres"].data_ptr())

    def step():
        for c in chunks:
            run_chunk(c, d_cig)
        for b in swb:  # the step's mate rescue: the pairs of all chunks, in batches large enough to fill the chip
            run_sw(b)

    def sync_all():
        for s in streams:
            s.synchronize()

    def barrier():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()

    note(f"uploaded in {upload_s:.1f} s; warm-up")
    for _ in range(args.warmup):
        step()
    barrier()
    note("timing")
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t_start
    for c in ctxs:
        c.sync()  # surfaces any BMH_E_RANGE / BMH_E_CIGAR_CAP flagged by a kernel
    if use_dist:
        dist.barrier()
    rank_ms = elapsed / args.steps * 1e3
    elapsed, reads_all, tasks_all = sh.reduce_report(elapsed, n_reads_step, n_seeded + n_glb + n_sw, rdev)
    per_rank_ms = [rank_ms]
    if use_dist:
        t = torch.zeros(world, dtype=torch.float64, device=rdev)
        t[rank] = rank_ms
        dist.all_reduce(t)
        per_rank_ms = [float(x) for x in t.cpu()]

    # ---- the same steps fed from pinned host memory over PCIe, results downloaded: a second timed region (`value_streamed`)
    feed = streamed = None
    if args.feed in ("host", "both"):
        note(f"timed: {rank_ms:.1f} ms per step resident; pinning host buffers for the host-fed steps")
        t0 = time.time()
        feed = HostFeed(torch, dev, pkg, tg, host, host_sw, (cx_exts, cx_glb, cx_sw), (ext_streams, streams[1 % len(streams)], streams[2 % len(streams)]), ship_global_windows=args.feed_global_windows)
        pin_s = time.time() - t0
        for _ in range(max(1, args.warmup)):
            feed.step()
        feed.drain(); sync_all(); barrier()
        t_start = time.perf_counter()
        for _ in range(args.steps):
            feed.step()
        sync_all(); feed.drain()
        torch.cuda.synchronize(dev)
        el_s = time.perf_counter() - t_start
        if use_dist:
            dist.barrier()
        rank_ms_s = el_s / args.steps * 1e3
        el_s_all, _, _ = sh.reduce_report(el_s, n_reads_step, 0, rdev)
        feed.step(timed=True)  # one more, with HIP events around every chunk's upload on the copy stream
        sync_all(); feed.drain()
        up_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in feed.copy_events)
        up_bytes = sum(b for _, _, b in feed.copy_events)
        streamed = {"value_streamed": reads_all * args.steps / el_s_all, "ms_per_step": el_s_all / args.steps * 1e3, "rank_ms_per_step": rank_ms_s,
                    "h2d_bytes_per_step": feed.h2d_bytes, "d2h_bytes_per_step": feed.d2h_bytes, "resident_reference_window_bytes": feed.resident_window_bytes,
                    "h2d_GBps_copy_stream": up_bytes / (up_ms * 1e-3) / 1e9 if up_ms > 0 else None,
                    "pcie_GBps_over_the_step": (feed.h2d_bytes + feed.d2h_bytes) / (rank_ms_s * 1e-3) / 1e9,
                    "pin_s": pin_s,
                    "global_windows_shipped": bool(args.feed_global_windows),
                    "what": "reads + task records (+ the global tasks' query bytes; their target windows stay resident like the seeds', as with the "
                            "library's region record -- unless --feed-global-windows) in pinned host memory, each chunk uploaded one chunk "
                            "ahead of compute (copy stream), kernels as in the resident step, finished regions / global scores / CIGAR words "
                            "(16 words of room per task) / rescue results downloaded to pinned host memory (third stream), all inside the timed "
                            "region; the seed records' reference windows are resident in HBM, as the reference is in the preload shim"}

    note(f"timed: {rank_ms:.1f} ms per step" + (f", host-fed {streamed['rank_ms_per_step']:.1f} ms" if streamed else "") + "; instrumented pass")
    # ---- one more, instrumented pass (untimed): HIP events on the launch streams around every stage and every kernel
    # of the dominant stage, chunk by chunk
    stage_ms = {"seed_extension": 0.0, "global_alignment": 0.0, "mate_rescue_sw": 0.0}
    round_ms = np.zeros(4)
    gbin_ms = np.zeros(3)
    for c in ctxs:
        c.set_kernel_timing(True)
    cx_ext.extend_bin_ms_sum(reset=True)
    evs = []
    for c in chunks:
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        e[0].record(streams[0]); cx_ext.seedext_batch_device(c["pool"].data_ptr(), c["seeds"].data_ptr(), c["n"], c["sres"].data_ptr()); e[1].record(streams[0])
        sg = streams[1 % len(streams)]
        e[2].record(sg); cx_glb.global_batch_device(c["gpool"].data_ptr(), c["gtasks"].data_ptr(), c["ng"], c["gres"].data_ptr(), d_cig.data_ptr()); e[3].record(sg)
        sync_all()
        round_ms += np.array(cx_ext.last_seedext_round_ms())
        gbin_ms += np.maximum(np.array(cx_glb.last_global_bin_ms()), 0.0)
        evs.append(e)
    for e in evs:
        stage_ms["seed_extension"] += e[0].elapsed_time(e[1])
        stage_ms["global_alignment"] += e[2].elapsed_time(e[3])
    ss = streams[2 % len(streams)]
    for b in swb:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(ss); run_sw(b); e1.record(ss)
        sync_all()
        stage_ms["mate_rescue_sw"] += e0.elapsed_time(e1)
    ext_bin_ms, ext_bin_launches = cx_ext.extend_bin_ms_sum(reset=True)  # per extension kernel, summed over the fused rounds of all chunks
    for c in ctxs:
        c.set_kernel_timing(False)

    out = None
    if rank == 0:
        note("parity check and CPU baseline")
        # ---- parity on chunk 0 (its results are still on the device; CIGARs: run its global batch once more) + CPU baseline
        c0, h0 = chunks[0], host[0]
        pool, seeds, gpool, gtasks, gwords = h0
        spool, stasks = host_sw[0]
        cig0 = torch.zeros(gwords + 8, dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)  # the fill runs on torch's current stream, the kernels on the context's: without this the fill can overtake the kernel's first CIGAR words
        cx_glb.global_batch_device(c0["gpool"].data_ptr(), c0["gtasks"].data_ptr(), c0["ng"], c0["gres"].data_ptr(), cig0.data_ptr())
        sync_all()
        sres = c0["sres"].cpu().numpy().view(pkg.SEED_RES)
        gres = c0["gres"].cpu().numpy().view(pkg.GLB_RES)
        gcig = cig0.cpu().numpy().view(np.uint32)
        swres = swb[0]["swres"].cpu().numpy().view(pkg.SW_RES)
        ns = min(len(seeds), 20000)
        want, cells, calls = kswlib.orc_seedext_batch(params, pool, seeds[:ns], nthreads=ncores)
        ok_ext = all(bool((want[f] == sres[:ns][f]).all()) for f in want.dtype.names)
        cells_per_seed, calls_per_seed = cells / ns, calls / ns
        ng = min(len(gtasks), 5000)
        ores, ocig, _ = kswlib.orc_global_batch_mt(params, gpool, gtasks[:ng], gwords, nthreads=ncores)
        ok_glb = bool((ores == gres[:ng]).all())
        for k in range(0, ng, 7):
            o, nn = int(gtasks[k]["cigar_off"]), int(ores[k]["n_cigar"])
            ok_glb = ok_glb and bool((ocig[o:o + nn] == gcig[o:o + nn]).all())
        nw = min(len(stasks), 3000)
        swant, _ = kswlib.orc_sw_batch(params, spool, stasks[:nw], nthreads=ncores)
        ok_sw = all(bool((swant[f] == swres[:nw][f]).all()) for f in kswlib.SW_FIELDS)
        ok_feed = True
        if feed is not None:  # what the host-fed steps brought back == what the resident steps left on the device
            fo = feed.chunks[0]["out"]
            ok_parts = [bool((fo[0].numpy().view(pkg.SEED_RES) == sres).all()), bool((fo[1].numpy().view(pkg.GLB_RES) == gres).all()),
                        bool((feed.sw[0]["out"].numpy().view(pkg.SW_RES) == swres).all())]
            ok_feed = all(ok_parts)
            if not ok_feed:
                note(f"host-fed results differ from the resident ones: seed records {ok_parts[0]}, global results {ok_parts[1]}, rescue results {ok_parts[2]}")
            fc = fo[2].numpy().view(np.uint32)
            n_long = 0
            for k in range(0, len(gtasks), 11):
                o, nn = int(gtasks[k]["cigar_off"]), int(gres[k]["n_cigar"])
                if nn > HostFeed.CIGAR_WORDS:  # the driver would redo this one with worst-case room (counted, not compared)
                    n_long += 1
                    continue
                fo_ = k * HostFeed.CIGAR_WORDS
                same = bool((fc[fo_:fo_ + nn] == gcig[o:o + nn]).all())
                if not same and ok_feed:
                    note(f"host-fed CIGAR of global task {k} differs: {fc[fo_:fo_ + nn][:8]} vs {gcig[o:o + nn][:8]} (n_cigar {nn})")
                ok_feed = ok_feed and same
            streamed["cigars_longer_than_16_words_in_sample"] = n_long
        parity_ok = ok_ext and ok_glb and ok_sw  # `value`: the resident steps against the oracle; the host-fed steps have their own verdict

        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            # the oracle port replaying the SAME records on all host threads: K reads' worth of each stage, sized from a
            # first pass so that the whole sample costs about --cpu-seconds
            t1 = time.perf_counter()
            kswlib.orc_seedext_batch(params, pool, seeds[:ns], nthreads=ncores)
            kswlib.orc_global_batch_mt(params, gpool, gtasks[:ng], gwords, nthreads=ncores)
            kswlib.orc_sw_batch(params, spool, stasks[:nw], nthreads=ncores)
            probe = time.perf_counter() - t1
            per_read = probe / (ns / (len(seeds) / sizes[0]))  # crude seconds per read of the mix
            K = int(min(sizes[0], max(20000, args.cpu_seconds / max(per_read, 1e-9) * 0.7)))
            ke = min(len(seeds), int(K * len(seeds) / sizes[0]))
            kg = min(len(gtasks), int(K * len(gtasks) / sizes[0]))
            kw = max(1, min(len(stasks), int(K * n_sw / n_reads_step)))
            t1 = time.perf_counter()
            w1, cells1, calls1 = kswlib.orc_seedext_batch(params, pool, seeds[:ke], nthreads=ncores)
            t_e = time.perf_counter() - t1
            t1 = time.perf_counter()
            g1, gc1, _ = kswlib.orc_global_batch_mt(params, gpool, gtasks[:kg], gwords, nthreads=ncores)
            t_g = time.perf_counter() - t1
            t1 = time.perf_counter()
            s1, _ = kswlib.orc_sw_batch(params, spool, stasks[:kw], nthreads=ncores)
            t_s = time.perf_counter() - t1
            parity_ok = parity_ok and all(bool((w1[f] == sres[:ke][f]).all()) for f in w1.dtype.names) and bool((g1 == gres[:kg]).all()) \
                and all(bool((s1[f] == swres[:kw][f]).all()) for f in kswlib.SW_FIELDS)
            cells_per_seed, calls_per_seed = cells1 / ke, calls1 / ke
            cpu = {"value": K / (t_e + t_g + t_s), "unit": "reads/s", "cores": ncores, "kind": "port",
                   "sample": f"the first {K} reads' worth of chunk 0 of the same step ({ke} fused seed extensions, {kg} global alignments, "
                             f"{kw} mate-rescue Smith-Watermans) replayed by oracle/*.c on {ncores} pthreads: "
                             f"{t_e:.2f} + {t_g:.2f} + {t_s:.2f} s; every result compared with the GPU's",
                   "cores_note": cores_note,
                   "stage_seconds": {"seed_extension": t_e, "global_alignment": t_g, "mate_rescue_sw": t_s}}

        ms_per_step = elapsed / args.steps * 1e3
        value = reads_all * args.steps / elapsed
        # ---- kernels of the step (instrumented pass): the fused extension's four rounds, the three global kernels, the SW stage
        ql, tl, gw = (np.concatenate([h[3][f].astype(np.int64) for h in host]) for f in ("qlen", "tlen", "w"))
        worst = int(params["o_del"]) + int(params["o_ins"]) + max(int(params["e_del"]), int(params["e_ins"])) * (ql + tl) + \
            max(int(-params["mat"].min()), int(params["mat"].max())) * np.maximum(ql, tl)
        lane_ok = (tl <= 512) & (worst < 12000)
        gbin = np.where(lane_ok & (gw <= 31), 0, np.where(lane_ok & (gw <= 63), 1, 2))
        g_ncig_mean = float(gres["n_cigar"].mean())
        g_bytes = ql + tl + 40 + 4 * g_ncig_mean  # qlen + tlen + 32 + 8 + 4*n_cigar
        seeds_all_q = np.concatenate([h[1]["l_query"].astype(np.int64) for h in host])
        seeds_all_w = np.concatenate([h[1]["wlen"].astype(np.int64) for h in host])
        ext_bytes = float((seeds_all_q + seeds_all_w).sum() + n_seeded * (40 + 32))  # read + window + the 40 B record + 32 B result
        sw_bytes = float(sum((t["qlen"].astype(np.int64) + t["tlen"].astype(np.int64)).sum() + 64 * len(t) for _, t in host_sw))
        kernels = []
        names = ["round L1: left extensions (extend_lane_kernel<32|64|128> + seed_left_make)", "round L2: left retries at 2w",
                 "round R1: right extensions, h0 = the device's left score", "round R2: right retries at 2w"]
        for k in range(4):
            kernels.append({"kernel": "seedext " + names[k], "ms": float(round_ms[k]), "launches": n_chunks, "single_kernel": False,
                            "algorithmic_bytes": ext_bytes / 2 if k in (0, 2) else 0.0})
        # the extension kernels themselves (a round is several of them): time per length bin summed over the rounds, HIP events on the
        # launch streams; the algorithmic bytes of the fused records are split over the bins in proportion to their time
        ext_names = ["extend_lane_kernel<32, true, false>", "extend_lane_kernel<64, true, false>", "extend_lane_kernel<96, true, false> + <128, true, false> (65-128 columns)",
                     "extend_lanex_kernel<2> / extend_reg_kernel<4> (129-256 columns)", "extend_lanex_kernel<4> (257-512 columns)", "extend_lds_kernel (longer)"]
        ext_tot = sum(ext_bin_ms) or 1.0
        for b in range(6):
            if ext_bin_ms[b] > 0:
                kernels.append({"kernel": ext_names[b], "ms": float(ext_bin_ms[b]), "launches": max(1, ext_bin_launches), "single_kernel": b < 2,
                                "algorithmic_bytes": ext_bytes * ext_bin_ms[b] / ext_tot})
        gnames = ["global_lane_kernel<64, true, true> (ksw_global2 + traceback, w <= 31, 64 tasks/wave)", "global_lane_kernel<96> + <128> (32 <= w <= 47, 48 <= w <= 63)",
                  "global_kernel (one wave per task: wide bands, long targets)"]
        for b in range(3):
            kernels.append({"kernel": gnames[b], "ms": float(gbin_ms[b]), "launches": n_chunks, "tasks": int((gbin == b).sum()), "single_kernel": b != 1,
                            "algorithmic_bytes": float(g_bytes[gbin == b].sum())})
        kernels.append({"kernel": "sw_lane_kernel<80, true, false, false> + second pass (ksw_align2, mate rescue)", "ms": stage_ms["mate_rescue_sw"], "launches": n_swb,
                        "tasks": n_sw, "algorithmic_bytes": sw_bytes, "single_kernel": False})
        for k in kernels:
            k["avg_launch_ms"] = k["ms"] / k["launches"]
            k["GBps"] = k["algorithmic_bytes"] / (k["ms"] * 1e-3) / 1e9 if k["ms"] > 0 else None
        # the dominant kernel of the roofline object: the ONE kernel (a row of the rocprofv3 summary in profiles/) with the largest total
        # time in the step -- whichever family it belongs to
        dom = max((k for k in kernels if k["single_kernel"]), key=lambda k: k["ms"])
        dom_name = dom["kernel"].split(" (")[0]
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from csrc_sha import csrc_sha
        tree_sha = csrc_sha(ROOT)
        traffic = traffic_src = valu = None
        insts_per_launch = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):  # HBM bytes and VALU instructions per launch from rocprofv3 PMC passes of this same command (tools/profile_bench.sh)
            tj = json.load(open(tpath))
            if tj.get("csrc_sha") != tree_sha:
                traffic_src = (f"profiles/traffic_latest.json was produced from kernel sources {tj.get('csrc_sha', '(unstamped)')}, this tree is {tree_sha}: "
                               f"not quoted (re-run tools/profile_bench.sh + tools/summarize_prof.py)")
            else:
                for kname, v in tj.get("kernels", {}).items():
                    if kname.replace("bmh::", "").startswith(dom_name):
                        traffic = v.get("hbm_bytes_per_launch")
                        insts_per_launch = v.get("valu_insts_per_launch")
                        traffic_src = f"profiles/traffic_latest.json (kernel sources {tree_sha}; rocprofv3 FETCH_SIZE+WRITE_SIZE and SQ_INSTS_VALU passes of this command, {tj.get('source')})"
        # what actually bounds these kernels: the vector-instruction issue rate, which on gfx950 depends on the instruction class
        # (profiles/r03_valu_issue_classes.md: 2 cycles per SIMD for the fast class, 4 for the slow class, measured).  The kernel's own
        # mix (profiles/r03_valu_mix.json, static count over its DP row loop, same source hash) gives its mix-weighted peak.
        mpath = os.path.join(ROOT, "profiles", "r03_valu_mix.json")
        if os.path.exists(mpath):
            mj = json.load(open(mpath))
            mk = mj.get("kernels", {}).get(dom_name)
            if mk and mj.get("csrc_sha") == tree_sha:
                peak_mix = mk["peak_mix_weighted_Ginst_s"]
                ach = insts_per_launch / (dom["avg_launch_ms"] * 1e-3) / 1e9 if insts_per_launch else None
                valu = {"bound": "valu-issue", "kernel": dom_name, "mix": {"fast_2cyc": mk["fast"], "slow_4cyc": mk["slow"], "slow_8cyc": mk["slow8"]},
                        "cycles_per_instruction_mix_weighted": mk["cycles_per_valu"], "peak_mix_weighted": peak_mix,
                        "peak_all_slow": 256 * 4 * mj["clock_GHz"] / 4, "peak_all_fast": 256 * 4 * mj["clock_GHz"] / 2,
                        "achieved": ach, "unit": "G wave-instructions/s", "frac": ach / peak_mix if ach else None,
                        "valu_insts_per_launch": insts_per_launch, "valu_per_cell_static": mk.get("valu_per_cell"),
                        "source": "profiles/r03_valu_issue_classes.md (measured classes), profiles/r03_valu_mix.json (this kernel's loop), "
                                  "SQ_INSTS_VALU from " + (traffic_src or "no PMC file for this tree"),
                        "note": "the launch shares the chip with the other stages' kernels (3 streams); the class costs hold from 2 resident waves per SIMD on "
                                "in a stream that is not pure vector code (profiles/r03_valu_pairing.md, DESIGN.md 4.2)"}
            elif mk:
                valu = {"skipped": f"profiles/r03_valu_mix.json is from kernel sources {mj.get('csrc_sha')}, this tree is {tree_sha} (python tools/make_valu_mix_profile.py)"}
        metric = "aligned reads/sec (150 bp PE vs hg38) at 1/2/4/8 MI355X; SAM bit-exact vs CPU"
        bpath = os.path.join(ROOT, "BASELINE.json")
        if os.path.exists(bpath):
            metric = json.load(open(bpath)).get("metric", metric)
        step_cells = cells_per_seed * n_seeded
        out = {
            "metric": metric,
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int16", "data": "synthetic",
            "config": {"workload": f"pe10m -- BASELINE.json configs[2]: {args.pairs} x 2 x {args.shape} read pairs per GPU per step through the whole DP path "
                                   f"(fused seed extension with the right extension started from the device's left score, ksw_global2 + traceback, "
                                   f"mate-rescue ksw_align2), task generator in place of the hg38 index (SURVEY.md §8d), streamed in {n_chunks} chunks, "
                                   f"inputs resident in HBM; seeding, chaining and SAM text are host stages outside the step (see cpu_baseline_pipeline)",
                       "pairs_per_gpu": args.pairs, "reads_per_gpu": n_reads_step, "chunks": n_chunks, "seeded_reads_per_gpu": n_seeded,
                       "ksw_extend2_calls_per_gpu": int(round(calls_per_seed * n_seeded)), "global_tasks_per_gpu": n_glb, "rescue_tasks_per_gpu": n_sw,
                       "rescue_batches": n_swb, "mean_global_w": float(gw.mean()), "streams": args.streams,
                       "parallelism": f"static shard x{world}, one process per GPU, no collective on the data path"},
            "tasks_per_s": tasks_all * args.steps / elapsed,
            "value_streamed": streamed["value_streamed"] if streamed else None,
            "streamed": streamed,
            "per_rank_ms_per_step": per_rank_ms,
            "parity": ("bit-exact vs oracle: fused seed extensions, global alignments (scores, n_cigar, sampled CIGARs), mate-rescue SW"
                       if parity_ok else f"MISMATCH vs oracle (ext {ok_ext}, global {ok_glb}, sw {ok_sw})"),
            "parity_streamed": (None if feed is None else "host-fed results identical to the resident ones (finished regions, global scores, every 11th CIGAR, rescue results)"
                                if ok_feed else "MISMATCH between host-fed and resident results"),
            "stages_ms_per_step": {k: v for k, v in stage_ms.items()},
            "stage_rates": {"seed_extension_gcups": step_cells / (stage_ms["seed_extension"] * 1e-3) / 1e9,
                            "ksw_extend2_per_s": calls_per_seed * n_seeded / (stage_ms["seed_extension"] * 1e-3),
                            "global_tasks_per_s": n_glb / (stage_ms["global_alignment"] * 1e-3),
                            "rescue_tasks_per_s": n_sw / (stage_ms["mate_rescue_sw"] * 1e-3)},
            "roofline": {"bound": "hbm", "bound_measured": "valu-issue (integer max-plus DP, no MFMA form)", "achieved": dom["GBps"], "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": (dom["GBps"] or 0.0) / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": dom["kernel"], "kernel_ms": dom["avg_launch_ms"], "valu_issue": valu,
                         "algorithmic_bytes_per_launch": dom["algorithmic_bytes"] / dom["launches"], "kernels": kernels,
                         "note": "achieved = algorithmic bytes of the dominant kernel's tasks / its HIP-event duration (instrumented pass, "
                                 "events on the launch stream); integer DP is VALU-issue bound, GCUPS / tasks/s are the honest figures"},
            "cpu_baseline": cpu,
            "cpu_baseline_pipeline": pipe,
            "setup": {"taskgen_s": gen_s, "upload_s": upload_s, "host_threads_for_generation": workers},
        }
        # what `value` covers, so that it cannot be read as an end-to-end figure: the DP path only, on generator tasks
        out["value_scope"] = ("DP-path reads/s (seed extension + global alignment + mate rescue) on synthetic generator tasks, inputs resident in HBM; "
                              "value_streamed = the same steps fed over PCIe from pinned host memory with results downloaded; "
                              "value_end_to_end = whole `bwa mem` (seeding, chaining, DP, SAM text) through the preload shim, SAM identical to the reference")
        out["value_end_to_end"] = (pipe or {}).get("dut_value") if (pipe or {}).get("sam_identical") else None
        if not parity_ok:
            out["value"] = 0.0  # a fast kernel with different results is not done
        if streamed and not (parity_ok and ok_feed):
            out["value_streamed"] = 0.0
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    sys.exit(main() or 0)
