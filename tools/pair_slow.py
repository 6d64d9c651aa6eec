#!/usr/bin/env python3
"""pair_slow.py -- build step: gives every 4-cycle-class vector instruction of a gfx950 assembly file a non-VALU instruction next to it.

    python3 tools/pair_slow.py in.s out.s [--stats]

Why (measured: tools/microbench/valu_dep.hip, profiles/r03_valu_pairing.md): on gfx950 a vector instruction of the 2-cycle class (v_add_u32,
16-bit VOP2, v_bitop3, shifts right ...) issues every 2 cycles per SIMD and one of the 4-cycle class (v_max_i32, v_perm, v_lshl_or, every
SDWA/DPP/VOP3P form, v_cmp/v_cndmask ...) every 4 -- but only while each 4-cycle instruction has a NON-VALU instruction (s_nop 0 or any SALU
instruction) directly before or behind it in the wave's stream.  A 4-cycle instruction with vector instructions on both sides makes the
whole stream cost 4 cycles per instruction: 1 slow + 7 fast instructions take 32 cycles instead of 18, with an `s_nop 0` behind the slow
one 18.3.  hipcc knows nothing of this, so the DP kernels' device assembly passes through here between `hipcc -S` and the assembler
(bwa-mem-quickassist_amd/Makefile): behind every slow-class instruction that has no free neighbour an `s_nop 0` is inserted.  Nothing else is
touched; an s_nop is legal anywhere in a wave's stream (the one place where adjacency matters, s_getpc_b64 + s_add_u32, has no vector
instruction in it).
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from valu_mix import classify  # noqa: E402  (the instruction classes, as measured)

NOT_FREE = ("s_waitcnt", "s_cbranch", "s_branch", "s_load", "s_buffer_load", "s_store", "s_barrier", "s_endpgm", "s_sendmsg", "s_sleep", "s_dcache",
            "s_icache", "s_getpc", "s_setpc", "s_swappc", "s_code_end", "s_trap", "s_setprio", "s_sethalt", "s_setreg", "s_getreg", "s_memtime",
            "s_memrealtime", "s_atomic", "s_scratch", "s_set_gpr", "s_cbranch_g", "s_call", "s_rfe", "s_inst_prefetch", "s_clause", "s_ttrace")
INSTR = re.compile(r"^\s+([a-z_0-9]+)\b(.*)$")


def kind(line):
    """'slow', 'free' (an instruction that can sit next to a slow one), 'other' (any other instruction) or None (label, directive, comment)"""
    s = line.split(";")[0].rstrip() if not line.lstrip().startswith(";") else ""
    m = INSTR.match(s)
    if not m or s.lstrip().startswith("."):
        return None
    op, text = m.group(1), m.group(2)
    if op.startswith("s_"):
        return "other" if op.startswith(NOT_FREE) else "free"
    if op.startswith("v_"):
        return "slow" if classify(op, text) in ("slow", "slow8") else "other"
    return "other"


def pair(lines):
    """-> (new lines, slow instructions seen, nops inserted).  Labels end adjacency (control may arrive from elsewhere)."""
    out, n_slow, n_ins = [], 0, 0
    kinds = [kind(l) for l in lines]
    used = [False] * len(lines)  # free instructions already standing next to a slow one
    # index of the previous / next INSTRUCTION with nothing but comments in between (labels and directives break the link)
    def neighbour(i, step):
        j = i + step
        while 0 <= j < len(lines):
            if kinds[j] is not None:
                return j
            t = lines[j].strip()
            if t and not t.startswith(";") and not t.startswith("//"):
                return -1  # a label or a directive
            j += step
        return -1
    for i, l in enumerate(lines):
        out.append(l)
        if kinds[i] != "slow":
            continue
        n_slow += 1
        p = neighbour(i, -1)
        if p >= 0 and kinds[p] == "free" and not used[p]:
            used[p] = True
            continue
        n = neighbour(i, +1)
        if n >= 0 and kinds[n] == "free" and not used[n]:
            used[n] = True
            continue
        out.append("\ts_nop 0 ; pair\n")
        n_ins += 1
    return out, n_slow, n_ins


def main():
    src, dst = sys.argv[1], sys.argv[2]
    lines = open(src).readlines()
    out, n_slow, n_ins = pair(lines)
    open(dst, "w").writelines(out)
    if "--stats" in sys.argv:
        print(f"{os.path.basename(src)}: {n_slow} slow-class instructions, {n_ins} s_nop 0 inserted", file=sys.stderr)


if __name__ == "__main__":
    main()
