#!/usr/bin/env python3
"""Determinism of the global-alignment batch under load: one quiet run as the reference, then runs with a fused-extension batch in
flight on another stream and with host<->device copies on two more; every CIGAR word of every task is compared.  (GPU box only.)"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import kswlib
from __graft_entry__ import load_package
pkg = load_package(); tg = importlib.import_module("bwa_mem_quickassist_amd.taskgen")
p = kswlib.make_params()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_400_000
gpool, gtasks, gwords = tg.generate_global(n, "150bp", seed=22)
spool, seeds = tg.generate_seeds(p, 4_000_000, "150bp", seed=3)
dev = torch.device("cuda", 0)
cg, ce = pkg.Context(0, p), pkg.Context(0, p)
sg, se, sc1, sc2 = (torch.cuda.Stream(dev) for _ in range(4))
cg.set_qcap(int(gtasks["qlen"].max())); cg.set_stream(sg.cuda_stream)
ce.set_qcap(160); ce.set_stream(se.cuda_stream)
up = lambda a: torch.from_numpy(a.view(np.uint8).reshape(-1)).to(dev)
d_pool, d_t, d_sp, d_sd = up(gpool), up(gtasks), up(spool), up(seeds)
d_sr = torch.zeros(len(seeds) * pkg.SEED_RES.itemsize, dtype=torch.uint8, device=dev)
big_h = torch.empty(1 << 30, dtype=torch.uint8).pin_memory(); big_d = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
off = gtasks["cigar_off"].astype(np.int64)
def run(load):
    d_res = torch.zeros(len(gtasks) * pkg.GLB_RES.itemsize, dtype=torch.uint8, device=dev)
    d_cig = torch.zeros(gwords + 8, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    if load:
        ce.seedext_batch_device(d_sp.data_ptr(), d_sd.data_ptr(), len(seeds), d_sr.data_ptr())
        with torch.cuda.stream(sc1): big_d.copy_(big_h, non_blocking=True)
        with torch.cuda.stream(sc2): big_h.copy_(big_d, non_blocking=True)
    cg.global_batch_device(d_pool.data_ptr(), d_t.data_ptr(), len(gtasks), d_res.data_ptr(), d_cig.data_ptr())
    if load:
        ce.seedext_batch_device(d_sp.data_ptr(), d_sd.data_ptr(), len(seeds), d_sr.data_ptr())
    torch.cuda.synchronize()
    return d_res.cpu().numpy().view(pkg.GLB_RES), d_cig.cpu().numpy().view(np.uint32)
r0, c0 = run(False)
nn = r0["n_cigar"].astype(np.int64)
idx = np.repeat(off, nn) + (np.arange(nn.sum()) - np.repeat(np.cumsum(nn) - nn, nn))  # every CIGAR word of every task
for rep in range(4):
    r, c = run(rep > 0)
    diff = c[idx] != c0[idx]
    bad_tasks = np.unique(np.repeat(np.arange(len(gtasks)), nn)[diff])
    print(f"run {rep} ({'under load' if rep else 'quiet'}): results equal {bool((r == r0).all())}, tasks with a different CIGAR word: {len(bad_tasks)}", bad_tasks[:6])
    for k in bad_tasks[:2]:
        print("   task", k, dict(zip(gtasks.dtype.names, gtasks[k])), "quiet", c0[off[k]:off[k] + nn[k]][:10], "now", c[off[k]:off[k] + nn[k]][:10])
