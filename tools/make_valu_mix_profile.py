#!/usr/bin/env python3
"""Instruction mix of the hot loops of the lane kernels, priced with the measured issue classes (tools/valu_mix.py), as
profiles/r03_valu_mix.json (read by bench.py; stamped with the kernel sources' hash) and profiles/r03_valu_mix.md.
Needs hipcc (cross-compiles without a GPU): python tools/make_valu_mix_profile.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import valu_mix
from csrc_sha import csrc_sha

KERNELS = [  # file, demangled-name pattern, DP cells one trip of the hot loop advances per lane
    ("extend_lane.hip", "extend_lane_kernel<32, true, false>", 32),
    ("extend_lane.hip", "extend_lane_kernel<64, true, false>", 64),
    ("extend_lane.hip", "extend_lane_kernel<96, true, false>", 96),
    ("extend_lane.hip", "extend_lane_kernel<128, true, false>", 128),
    ("global_lane.hip", "global_lane_kernel<64, true, true>", 128),   # (the loop holds the masked and the unmasked body of every block: 2 x C cells)
    ("global_lane.hip", "global_lane_kernel<96, true, true>", 192),
    ("global_lane.hip", "global_lane_kernel<128, true, true>", 256),
    ("sw_lane.hip", "sw_lane_kernel<80, true, false, false>", 160),
]
out = {"csrc_sha": csrc_sha(ROOT), "clock_GHz": valu_mix.CLOCK_GHZ, "cycles": valu_mix.CYC, "kernels": {}}
md = ["# Instruction mix of the lane kernels' DP row loops, priced with the measured issue classes (round 3)", "",
      "`python tools/make_valu_mix_profile.py` (static count over the hot loop of the gfx950 assembly; classes and their cost: "
      "`profiles/r03_valu_issue_classes.md`, `profiles/r03_valu_pairing.md`). The row loop of `global_lane_kernel<C, true, true>` holds two bodies of every "
      "block (masked and unmasked), of which a row runs one: its per-cell figures are the mean of the two.", "",
      "| kernel | VALU in loop | fast (2 cyc) | slow (4 cyc) | slow8 | cycles per VALU | mix-weighted issue peak, G wave-instr/s | VALU per cell | issue cycles per cell |",
      "|---|---|---|---|---|---|---|---|---|"]
for f, pat, cells in KERNELS:
    rs = valu_mix.report(os.path.join(ROOT, "bwa-mem-quickassist_amd", "csrc", f), pat, (), cells)
    if not rs:
        continue
    r = rs[0]
    c = r["counts"]
    out["kernels"][pat] = {"valu": r["valu_insts"], "fast": c.get("fast", 0), "slow": c.get("slow", 0), "slow8": c.get("slow8", 0),
                           "salu": c.get("salu", 0), "scratch": c.get("scratch", 0), "lds": c.get("lds", 0), "vmem": c.get("vmem", 0),
                           "cycles_per_valu": r["cycles_per_valu"], "peak_mix_weighted_Ginst_s": r["peak_mix_weighted_Ginst_s"],
                           "valu_per_cell": r.get("valu_per_cell"), "cycles_per_cell": r.get("cycles_per_cell"),
                           "top_slow": r["top_slow"][:8], "top_fast": r["top_fast"][:8]}
    md.append(f"| `{pat}` | {r['valu_insts']} | {c.get('fast', 0)} | {c.get('slow', 0)} | {c.get('slow8', 0)} | {r['cycles_per_valu']:.2f} | "
              f"{r['peak_mix_weighted_Ginst_s']:.0f} | {r.get('valu_per_cell', 0):.1f} | {r.get('cycles_per_cell', 0):.1f} |")
    md.append("")
    md[-1] = md[-1]
md += ["", "Most frequent instructions per kernel (count in the loop):", ""]
for pat, k in out["kernels"].items():
    md.append(f"* `{pat}` -- slow: " + ", ".join(f"{a} {b}" for a, b in k["top_slow"]) + "; fast: " + ", ".join(f"{a} {b}" for a, b in k["top_fast"]) +
              f"; scalar {k['salu']}, scratch {k['scratch']}, LDS {k['lds']}, global {k['vmem']}")
md += ["", f"Kernel sources: csrc hash `{out['csrc_sha']}`.  All-slow code peaks at {valu_mix.SIMDS * valu_mix.CLOCK_GHZ / 4:.0f} G wave-instructions/s, "
       f"all-fast code at {valu_mix.SIMDS * valu_mix.CLOCK_GHZ / 2:.0f} (1 024 SIMDs x {valu_mix.CLOCK_GHZ} GHz / cycles)."]
json.dump(out, open(os.path.join(ROOT, "profiles", "r03_valu_mix.json"), "w"), indent=1)
open(os.path.join(ROOT, "profiles", "r03_valu_mix.md"), "w").write("\n".join(l for l in md) + "\n")
print("\n".join(md))
