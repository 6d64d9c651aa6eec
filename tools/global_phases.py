#!/usr/bin/env python3
"""Where a global_lane launch spends its time: the same 150 bp task batch with CIGARs (fill + direction stores + traceback) and score-only
(cigar_cap = 0: no stores, no traceback).  Run on the GPU box: python tools/global_phases.py [tasks]"""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kswlib
from __graft_entry__ import load_package

pkg = load_package()
tg = importlib.import_module("bwa_mem_quickassist_amd.taskgen")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
p = kswlib.make_params()
gpool, gtasks, gwords = tg.generate_global(n, "150bp", seed=5)
dev = torch.device("cuda", 0)
ctx = pkg.Context(0, p)
ctx.set_qcap(int(gtasks["qlen"].max()))
s = torch.cuda.Stream(dev)
ctx.set_stream(s.cuda_stream)
up = lambda a: torch.from_numpy(a.view(np.uint8).reshape(-1)).to(dev)
d_pool = up(gpool)
d_res = torch.zeros(len(gtasks) * pkg.GLB_RES.itemsize, dtype=torch.uint8, device=dev)
d_cig = torch.zeros(gwords + 8, dtype=torch.int32, device=dev)
for name, mod in (("with CIGARs", None), ("score only (cigar_cap = 0)", 0)):
    t = gtasks.copy()
    if mod is not None:
        t["cigar_cap"] = 0
    d_t = up(t)
    for _ in range(2):
        ctx.global_batch_device(d_pool.data_ptr(), d_t.data_ptr(), len(t), d_res.data_ptr(), d_cig.data_ptr())
    s.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(3):
        ctx.global_batch_device(d_pool.data_ptr(), d_t.data_ptr(), len(t), d_res.data_ptr(), d_cig.data_ptr())
    e1.record(s)
    s.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"{name}: {ms:.2f} ms per {len(t)} tasks = {len(t) / ms / 1e3:.1f} M tasks/s")
ctx.close()
