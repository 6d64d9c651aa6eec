#!/usr/bin/env python3
"""valu_mix.py -- instruction mix of a kernel's hot loop, priced with the issue costs MEASURED on gfx950 (tools/microbench/valu_mix.hip,
profiles/r03_valu_issue_classes.md), and the mix-weighted issue peak that follows.

    python tools/valu_mix.py bwa-mem-quickassist_amd/csrc/extend_lane.hip 'extend_lane_kernel<128, true, false>' [--json]

The file is compiled to device assembly (hipcc -S --cuda-device-only, gfx950); the hot loop is the span closed by a backward branch that holds
the most vector instructions with (almost) no global loads in it and a plausible number of instructions per DP cell -- the unrolled DP
row loop, not the per-task set-up loop around it; every instruction in it is put into one of the classes below.

Measured (one SIMD, >= 2 resident waves, wall clock; cycles at the 2.39 GHz the chip holds under these loads):
  fast  2 cycles  v_add/sub/subrev_u32, and/or/xor/not/mov, lshrrev/ashrrev_b32, 16-bit VOP2 add/sub/max/min/shift, add/sub_u32|u16 clamp,
                  v_bitop3_b32, v_add_f32/fma/fmac, s_nop 0 (issue slot of the wave), v_accvgpr moves
  slow  4 cycles  max/min_i32|u32, lshlrev_b32, every v_cmp and v_cndmask, all SDWA and DPP forms, every other VOP3 (perm, alignbit/byte,
                  bfe/bfi, max3/min3/med3, add3, lshl_add, lshl_or, and_or, or3, mad_*24, mul), all VOP3P packed 16-bit, ffbh/ffbl/bcnt,
                  v_readlane/readfirstlane/writelane
  slow8 8 cycles  16-bit VOP3 three-operand forms (v_max3_i16, v_mad_u16, v_bitop3_b16, v_add/sub_i16 clamp)
Costs add (measured on mixes).  Scalar instructions issue beside the vector ones (no cost found in a mix); memory instructions are counted
but not priced: the loops are issue-bound, and their scratch/LDS/global traffic shows up as its own column.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

FAST = set("""v_add_u32 v_sub_u32 v_subrev_u32 v_and_b32 v_or_b32 v_xor_b32 v_xnor_b32 v_not_b32 v_mov_b32 v_lshrrev_b32 v_ashrrev_i32
v_add_u16 v_sub_u16 v_subrev_u16 v_max_u16 v_max_i16 v_min_u16 v_min_i16 v_lshlrev_b16 v_lshrrev_b16 v_ashrrev_i16 v_bitop3_b32
v_add_f32 v_sub_f32 v_mul_f32 v_fma_f32 v_fmac_f32 v_max_f16 v_accvgpr_read_b32 v_accvgpr_write_b32 v_accvgpr_mov_b32 v_nop
v_add_co_u32 v_addc_co_u32 v_sub_co_u32 v_subb_co_u32 v_subrev_co_u32""".split())
SLOW8 = set("v_max3_i16 v_max3_u16 v_min3_i16 v_min3_u16 v_med3_i16 v_med3_u16 v_mad_u16 v_mad_i16 v_bitop3_b16 v_add_i16 v_sub_i16".split())
CYC = {"fast": 2.0, "slow": 4.0, "slow8": 8.0}
CLOCK_GHZ = 2.39   # held under the slow-class loads of the microbenchmark (2.2-2.4 under fast-class ones)
SIMDS = 256 * 4


def classify(op, text):
    """class of one instruction line"""
    if op.startswith(("s_nop",)):
        return "fast"          # occupies the wave's issue slot like a 2-cycle instruction (measured)
    if op.startswith("s_"):
        return "salu" if not op.startswith(("s_waitcnt", "s_load", "s_buffer_load", "s_store", "s_dcache", "s_barrier", "s_sleep")) else "swait"
    if op.startswith(("scratch_",)):
        return "scratch"
    if op.startswith(("ds_",)):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "vmem"
    if not op.startswith("v_"):
        return "other"
    base = re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", op)
    if op.endswith(("_sdwa", "_dpp")) or " row_" in text or "quad_perm" in text or "dst_sel" in text:
        return "slow"
    if base in SLOW8:
        return "slow8"
    if base.startswith("v_cmp") or base.startswith("v_cndmask") or base.startswith("v_pk_"):
        return "slow"
    if base in FAST:
        return "fast"
    if base in ("v_add_u32", "v_sub_u32", "v_add_u16", "v_sub_u16") and "clamp" in text:
        return "fast"
    return "slow"


def assemble(hip, extra):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-S", "--cuda-device-only", "-I", os.path.join(root, "include"),
           "-mllvm", "-pragma-unroll-threshold=200000", "-o", out, hip] + extra
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names).encode(), capture_output=True)
    return p.stdout.decode().splitlines()


def kernels_of(asm):
    """{mangled: [lines]} for every kernel body of the .s file"""
    lines = open(asm).read().splitlines()
    out, cur = {}, None
    for l in lines:
        m = re.match(r"^(_Z\w+):\s", l)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is not None:
            out[cur].append(l)
            if l.strip().startswith("s_endpgm"):
                cur = None
    return out


def hot_loop(body, cells=None):
    """the unrolled DP row loop: of all spans closed by a backward branch the one with the most vector instructions that (a) has (almost)
    no global loads in it -- not the per-task set-up loop around it -- and (b), when the number of DP cells a trip advances is given,
    stays under 64 vector instructions per cell (larger spans take in the task loop or the traceback): (first, last) line indices"""
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    best = (0, len(body) - 1, -1)
    for i, l in enumerate(body):
        m = re.match(r"^\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            a = labels[m.group(1)]
            cnt, _ = mix_of(body[a:i + 1])
            valu = sum(cnt.get(c, 0) for c in CYC)
            if cnt.get("vmem", 0) * 100 <= valu and (not cells or valu <= 64 * cells) and valu > best[2]:
                best = (a, i, valu)
    return best[0], best[1]


def mix_of(lines):
    cnt, ops = {}, {}
    for l in lines:
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*)$", l)
        if not m or l.lstrip().startswith((";", ".")):
            continue
        op, text = m.group(1), m.group(2)
        c = classify(op, text)
        cnt[c] = cnt.get(c, 0) + 1
        key = re.sub(r"_(e32|e64)$", "", op)
        ops.setdefault(c, {})
        ops[c][key] = ops[c].get(key, 0) + 1
    return cnt, ops


def report(hip, pattern, extra=(), cells=None):
    asm = assemble(hip, list(extra))
    ks = kernels_of(asm)
    names = list(ks)
    dem = dict(zip(names, demangle(names)))
    res = []
    for n in names:
        d = dem[n]
        if pattern and pattern not in d:
            continue
        body = ks[n]
        a, b = hot_loop(body, cells)
        cnt, ops = mix_of(body[a:b + 1])
        valu = sum(cnt.get(c, 0) for c in CYC)
        cyc = sum(cnt.get(c, 0) * CYC[c] for c in CYC)
        r = {"kernel": d.split("(")[0], "loop_lines": b - a + 1, "counts": cnt, "valu_insts": valu, "valu_cycles": cyc,
             "cycles_per_valu": cyc / valu if valu else None,
             "peak_mix_weighted_Ginst_s": SIMDS * CLOCK_GHZ / (cyc / valu) if valu else None,
             "top_slow": sorted(ops.get("slow", {}).items(), key=lambda kv: -kv[1])[:14],
             "top_fast": sorted(ops.get("fast", {}).items(), key=lambda kv: -kv[1])[:10],
             "slow8": ops.get("slow8", {})}
        if cells:
            r["cells_in_loop"] = cells
            r["valu_per_cell"] = valu / cells
            r["cycles_per_cell"] = cyc / cells
        res.append(r)
    os.unlink(asm)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("hip")
    ap.add_argument("pattern", nargs="?", default="")
    ap.add_argument("--cells", type=int, default=0, help="DP cells one trip of the loop advances (per lane), for per-cell figures")
    ap.add_argument("--json", action="store_true")
    ap.add_argument("-D", action="append", default=[])
    a = ap.parse_args()
    res = report(a.hip, a.pattern, ["-D" + d for d in a.D], a.cells or None)
    if a.json:
        print(json.dumps(res))
        return
    for r in res:
        c = r["counts"]
        print(f"{r['kernel']}")
        print(f"  hot loop: {r['loop_lines']} lines; VALU {r['valu_insts']} (fast {c.get('fast', 0)}, slow {c.get('slow', 0)}, slow8 {c.get('slow8', 0)}), "
              f"SALU {c.get('salu', 0)}, scratch {c.get('scratch', 0)}, LDS {c.get('lds', 0)}, VMEM {c.get('vmem', 0)}, waits/smem {c.get('swait', 0)}")
        if r["valu_insts"]:
            print(f"  mix-weighted cost {r['cycles_per_valu']:.2f} cycles per VALU instruction -> issue peak {r['peak_mix_weighted_Ginst_s']:.0f} G wave-instructions/s "
                  f"(all-slow 612, all-fast 1224)")
        if "valu_per_cell" in r:
            print(f"  per cell: {r['valu_per_cell']:.1f} VALU, {r['cycles_per_cell']:.1f} issue cycles")
        print("  slow:", ", ".join(f"{k} {v}" for k, v in r["top_slow"]))
        print("  fast:", ", ".join(f"{k} {v}" for k, v in r["top_fast"]))
        if r["slow8"]:
            print("  slow8:", r["slow8"])


if __name__ == "__main__":
    main()
