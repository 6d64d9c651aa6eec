"""CPU checks of two pieces of tooling the measurements lean on: the pairing pass over device assembly (tools/pair_slow.py, the `make PAIR=1`
build of profiles/r03_valu_pairing.md) and the task generator's [queries | targets] split that bench.py's host-fed step uploads from."""
import os
import sys

import numpy as np

from __graft_entry__ import ROOT, load_package

sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_pairing_pass_gives_every_slow_instruction_a_free_neighbour():
    import pair_slow
    src = """\tv_add_u32_e32 v1, v1, v2
\tv_max_i32_e32 v3, v3, v2
\tv_add_u32_e32 v4, v4, v2
\ts_mov_b32 s4, 0
\tv_perm_b32 v5, v5, v2, v6
\tv_lshl_or_b32 v7, v7, 16, v2
\tv_sub_u16_e32 v8, v8, v2
.LBB0_1:
\tv_max3_i32 v9, v9, v2, v3
\ts_waitcnt lgkmcnt(0)
\tv_cndmask_b32_e32 v1, v1, v2, vcc
\ts_nop 0
\tv_pk_max_u16 v2, v2, v3
\ts_endpgm
""".splitlines(keepends=True)
    out, n_slow, n_ins = pair_slow.pair(src)
    text = "".join(out)
    assert n_slow == 6
    # v_max_i32: fast neighbours on both sides -> s_nop; v_perm: the s_mov before it serves; v_lshl_or: nothing free -> s_nop;
    # v_max3 behind a label, s_waitcnt does not count -> s_nop; v_cndmask and v_pk_max share ONE s_nop: only the first gets it
    assert n_ins == 4
    lines = [l.strip() for l in out]
    assert lines[lines.index("v_max_i32_e32 v3, v3, v2") + 1].startswith("s_nop 0")
    assert not lines[lines.index("v_perm_b32 v5, v5, v2, v6") + 1].startswith("s_nop")
    assert lines[lines.index("v_lshl_or_b32 v7, v7, 16, v2") + 1].startswith("s_nop 0")
    assert lines[lines.index("v_max3_i32 v9, v9, v2, v3") + 1].startswith("s_nop 0")
    assert lines[lines.index("v_pk_max_u16 v2, v2, v3") + 1].startswith("s_nop 0")
    # nothing but s_nop lines is added, and the original lines keep their order
    assert [l for l in out if "; pair" not in l] == src
    assert text.count("; pair") == n_ins


def test_global_task_split_keeps_every_sequence():
    load_package()
    import importlib
    tg = importlib.import_module("bwa_mem_quickassist_amd.taskgen")
    gpool, gtasks, _ = tg.generate_global(3000, "mixed100-300", seed=5)
    pool2, t2, qb = tg.split_queries_targets(gpool, gtasks)
    assert qb % 64 == 0 and len(t2) == len(gtasks)
    assert int(t2["qlen"].astype(np.int64).sum()) <= qb < int(t2["qlen"].astype(np.int64).sum()) + 64
    for k in range(0, len(gtasks), 37):
        a, b = gtasks[k], t2[k]
        assert (a["qlen"], a["tlen"], a["w"]) == (b["qlen"], b["tlen"], b["w"])
        assert int(b["q_off"]) + int(b["qlen"]) <= qb <= int(b["t_off"])
        assert np.array_equal(gpool[int(a["q_off"]):int(a["q_off"]) + int(a["qlen"])], pool2[int(b["q_off"]):int(b["q_off"]) + int(b["qlen"])])
        assert np.array_equal(gpool[int(a["t_off"]):int(a["t_off"]) + int(a["tlen"])], pool2[int(b["t_off"]):int(b["t_off"]) + int(b["tlen"])])
