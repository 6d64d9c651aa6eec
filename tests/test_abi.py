"""CPU-side checks of the boundary: the library builds/loads, exports every symbol the
header declares, record layouts match, and the no-GPU error path is loud (no fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import kswlib
from __graft_entry__ import load_package, build, ROOT


@pytest.fixture(scope="module")
def pkg():
    p = load_package()
    if not os.path.exists(p.LIB_PATH):
        build()
    return p


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.lib()
    names = pkg.declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/bwamem_hip.h but not exported"


def test_dropin_exports_reference_signatures(pkg):
    if not os.path.exists(pkg.DROPIN_PATH):
        build()
    out = os.popen(f"nm -D --defined-only {pkg.DROPIN_PATH}").read()
    for n in ("ksw_extend2", "ksw_global2", "ksw_align2", "ksw_align", "mem_align1_core_batched", "mem_align1_core", "mem_process_seqs"):
        assert re.search(rf"\bT {n}\b", out), n


def test_record_layouts_match_header(pkg):
    hdr = open(pkg.HEADER_PATH).read()
    assert "bmh_ext_task_t" in hdr and "bmh_glb_task_t" in hdr
    assert pkg.EXT_TASK.itemsize == 32 and pkg.EXT_RES.itemsize == 24
    assert pkg.GLB_TASK.itemsize == 32 and pkg.GLB_RES.itemsize == 8
    assert pkg.REGION_REQ.itemsize == 48 and pkg.REGION_RES.itemsize == 24
    assert pkg.PARAMS.itemsize == 64 and pkg.ALNREG.itemsize == 64 and pkg.SEED.itemsize == 16
    for a, b in ((pkg.EXT_TASK, kswlib.EXT_TASK), (pkg.EXT_RES, kswlib.EXT_RES), (pkg.GLB_TASK, kswlib.GLB_TASK)):
        assert a == b


def test_version_and_strerror(pkg):
    lib = pkg.lib()
    assert lib.bmh_version() == 310
    assert lib.bmh_strerror(0) == b"ok"
    assert b"range" in lib.bmh_strerror(pkg.BMH_E_RANGE)


def test_no_gpu_is_a_loud_error_not_a_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.BmhError) as e:
        pkg.Context(0, kswlib.make_params())
    assert e.value.code == pkg.BMH_E_NODEVICE


def test_product_does_not_link_the_oracle(pkg):
    """The shipped libraries must not depend on oracle/ (SURVEY §8c rule)."""
    for path in (pkg.LIB_PATH, pkg.DROPIN_PATH):
        if os.path.exists(path):
            out = os.popen(f"ldd {path}").read() + os.popen(f"nm -D {path}").read()
            assert "liborc" not in out and "orc_extend" not in out and "orc_global" not in out
