import sys, time, importlib
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, kswlib
from __graft_entry__ import load_package
pkg = load_package()
tg = importlib.import_module("bwa_mem_quickassist_amd.taskgen")
p = kswlib.make_params()
ctx = pkg.Context(0, p)
def t(f, n=6):
    f(); f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
for ng in (8192, 66000):
    gpool, gtasks, gwords = tg.generate_global(ng, "150bp", seed=3)
    gtasks = gtasks.copy(); gtasks["cigar_cap"] = 24; gtasks["cigar_off"] = np.arange(len(gtasks), dtype=np.uint32) * 24
    print("global_batch host-buffer", ng, "tasks: %.2f ms" % t(lambda: ctx.global_batch(gpool, gtasks, 24 * len(gtasks))), flush=True)
for ns in (8192, 32768):
    pool, seeds = tg.generate_seeds(p, ns, "150bp", seed=5)
    print("seedext_batch host-buffer", ns, "seeds: %.2f ms" % t(lambda: ctx.seedext_batch(pool, seeds)), flush=True)
for nw in (16, 200, 1500, 12000):
    spool, stasks = tg.generate_sw(p, nw, "150bp", seed=9)
    print("sw_batch host-buffer", nw, "tasks: %.2f ms" % t(lambda: ctx.sw_batch(spool, stasks)), flush=True)

# ---- the same calls from several host threads at once, one context each (what the preload shim does)
import threading
def conc(nthreads, make_call, label):
    ctxs = [pkg.Context(0, p) for _ in range(nthreads)]
    calls = [make_call(c) for c in ctxs]
    for f in calls: f(); f()
    times = [0.0] * nthreads
    def run(k):
        t0 = time.perf_counter()
        for _ in range(5): calls[k]()
        times[k] = (time.perf_counter() - t0) / 5 * 1e3
    th = [threading.Thread(target=run, args=(k,)) for k in range(nthreads)]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    wall = (time.perf_counter() - t0) / 5 * 1e3
    print(f"{label}: {nthreads} threads, per call {min(times):.2f}-{max(times):.2f} ms, wall per round {wall:.2f} ms", flush=True)
    for c in ctxs: c.close()
gpool, gtasks, gwords = tg.generate_global(66000, "150bp", seed=3)
gtasks = gtasks.copy(); gtasks["cigar_cap"] = 24; gtasks["cigar_off"] = np.arange(len(gtasks), dtype=np.uint32) * 24
pool, seeds = tg.generate_seeds(p, 32768, "150bp", seed=5)
spool, stasks = tg.generate_sw(p, 100, "150bp", seed=9)
for nt in (1, 4, 8):
    conc(nt, lambda c: (lambda: c.global_batch(gpool, gtasks, 24 * len(gtasks))), "global_batch 66k")
    conc(nt, lambda c: (lambda: c.seedext_batch(pool, seeds)), "seedext_batch 32k")
    conc(nt, lambda c: (lambda: c.sw_batch(spool, stasks)), "sw_batch 100")
