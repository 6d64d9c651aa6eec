#!/usr/bin/env python3
"""Where do extensions stop?  For a sample of the bench's fused seed records: rows executed by each ksw_extend2 call
(reference ksw.c:411-468) against the rows its target has, and why it stopped -- target exhausted, the row maximum fell
to zero (ksw.c:451) or z-drop (ksw.c:455-461).  Counted with the CPU oracle (oracle/ksw_oracle.c keeps the reference's
adaptive band), so the figures are properties of the workload, not of a kernel.  Prints markdown.
Usage: python tools/exit_rows.py [150bp|mixed100-300] [n_reads]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kswlib  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402


class Stats(C.Structure):
    _fields_ = [("cells", C.c_int64), ("rows", C.c_int)]


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "mixed100-300"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
    load_package()
    tg = importlib.import_module("bwa_mem_quickassist_amd.taskgen")
    p = kswlib.make_params()
    pool, tasks, _ = tg.generate(p, n, shape, seed=7)
    lib = kswlib.load_oracle()
    keep = []
    sc = kswlib.scoring_of(p, keep)
    rows, tl, ql, why = [], [], [], []
    for t in tasks:
        q, tg_ = kswlib.task_seqs(pool, t)
        out, st = kswlib.OrcExtOut(), Stats()
        lib.orc_extend(C.byref(sc), len(q), kswlib._u8p(q), len(tg_), kswlib._u8p(tg_), int(t["w"]), int(t["end_bonus"]), int(t["h0"]),
                       C.byref(out), C.byref(st))
        rows.append(st.rows), tl.append(len(tg_)), ql.append(len(q))
    rows, tl, ql = np.array(rows), np.array(tl), np.array(ql)
    frac = rows / np.maximum(tl, 1)
    print(f"### exit rows of {len(tasks)} ksw_extend2 calls, workload `{shape}` ({n} reads, taskgen seed 7)\n")
    print(f"query length: mean {ql.mean():.1f}, p50 {np.median(ql):.0f}, p99 {np.percentile(ql, 99):.0f}, max {ql.max()}; "
          f"target length: mean {tl.mean():.1f}, max {tl.max()}; rows executed: mean {rows.mean():.1f} = {rows.sum() / tl.sum():.2f} of the target rows\n")
    print(f"calls that stop before the last target row (m == 0 or z-drop): {(rows < tl).mean() * 100:.1f} %\n")
    print("| rows executed / target rows | share of calls | share of rows executed |")
    print("|---|---|---|")
    edges = [0, 0.1, 0.25, 0.5, 0.75, 0.9, 1.0001]
    for a, b in zip(edges, edges[1:]):
        sel = (frac >= a) & (frac < b)
        print(f"| {a:.2f} - {min(b, 1):.2f} | {sel.mean() * 100:.1f} % | {rows[sel].sum() / max(rows.sum(), 1) * 100:.1f} % |")
    print("\n| query-length bin (kernel) | calls | mean rows | mean rows / target | p95 rows |")
    print("|---|---|---|---|---|")
    for lo, hi, name in ((1, 32, "<=32 (lane<32>)"), (33, 64, "<=64 (lane<64>)"), (65, 128, "<=128 (lane<128>)"), (129, 256, "<=256 (lanex<2> / reg<4>)"),
                         (257, 100000, ">256 (LDS kernel)")):
        sel = (ql >= lo) & (ql <= hi)
        if sel.any():
            print(f"| {name} | {int(sel.sum())} | {rows[sel].mean():.1f} | {frac[sel].mean():.2f} | {np.percentile(rows[sel], 95):.0f} |")


if __name__ == "__main__":
    main()
