#!/usr/bin/env python3
"""ksw_align2 batches of mate-rescue shape (150 bp mate against its 450-850 bp window) at the sizes the preload shim sends:
one wave per task (sw_wave_kernel) against one lane per task (sw_lane_kernel, BMH_SW_WAVE=0), host buffers in and out.
Usage (GPU box): python tools/sw_small_batches.py"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kswlib  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
tg = importlib.import_module("bwa_mem_quickassist_amd.taskgen")
p = kswlib.make_params()
print("| tasks | one wave per task | one lane per task | same results |\n|---|---|---|---|")
for n in (64, 1000, 4000, 16000, 32000, 64000):
    pool, tasks = tg.generate_sw(p, n, "150bp", seed=13)
    ms, res = {}, {}
    for mode in ("1", "0"):
        os.environ["BMH_SW_WAVE"] = mode
        ctx = pkg.Context(0, p)
        ctx.sw_batch(pool, tasks)
        t0 = time.perf_counter()
        for _ in range(5):
            res[mode] = ctx.sw_batch(pool, tasks)
        ms[mode] = (time.perf_counter() - t0) / 5 * 1e3
        ctx.close()
    same = all((res["0"][f] == res["1"][f]).all() for f in kswlib.SW_FIELDS)
    note = "" if n <= 32768 else " (above 32 768 tasks both runs take the lane kernels)"
    print(f"| {n} | {ms['1']:.2f} ms | {ms['0']:.2f} ms{note} | {same} |", flush=True)
