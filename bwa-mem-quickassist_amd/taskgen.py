"""ctypes view of libbmh_taskgen.so (host/taskgen.c): synthetic extension workloads that
follow the task distribution mem_chain2aln produces.  Bench/test support."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
TASKGEN_PATH = os.path.join(_HERE, "libbmh_taskgen.so")


class Cfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("len_min", C.c_int32), ("len_max", C.c_int32),
                ("min_seed_len", C.c_int32), ("max_indel", C.c_int32), ("p_sub", C.c_double),
                ("p_ins", C.c_double), ("p_del", C.c_double), ("p_n", C.c_double), ("p_chimera", C.c_double)]


# BASELINE.json configs -> error models
WORKLOADS = {
    # C2-C4: 150 bp reads, 2 % subst + 0.25 % ins + 0.25 % del
    "150bp": dict(len_min=150, len_max=150, min_seed_len=19, max_indel=1, p_sub=0.02, p_ins=0.0025, p_del=0.0025,
                  p_n=0.0, p_chimera=0.0),
    # 2 x 250 bp (MiSeq-like): same error model; ksw_align2 runs in WORD mode from 250 columns on (bwamem_pair.c:147)
    "250bp": dict(len_min=250, len_max=250, min_seed_len=19, max_indel=1, p_sub=0.02, p_ins=0.0025, p_del=0.0025,
                  p_n=0.0, p_chimera=0.0),
    # C5: mixed 100-300 bp, indels 1-12, 30 % chimeric tails, 4 % N ("long-band stress")
    "mixed100-300": dict(len_min=100, len_max=300, min_seed_len=19, max_indel=12, p_sub=0.03, p_ins=0.005,
                         p_del=0.005, p_n=0.04, p_chimera=0.3),
}

_lib = None


def _load():
    global _lib
    if _lib is None:
        L = C.CDLL(TASKGEN_PATH)
        L.bmh_taskgen_ext.restype = C.c_int64
        L.bmh_taskgen_pool_bound.restype = C.c_size_t
        L.bmh_taskgen_glb.restype = C.c_int64
        _lib = L
    return _lib


def generate(params, n_reads, workload="150bp", seed=7, ext_task_dtype=None):
    """Returns (pool uint8[], tasks EXT_TASK[], task_read uint32[])."""
    from . import EXT_TASK, PARAMS
    L = _load()
    cfg = Cfg(seed=seed, **WORKLOADS[workload])
    p = np.ascontiguousarray(np.asarray(params, dtype=PARAMS).reshape(()))
    bound = L.bmh_taskgen_pool_bound(C.byref(cfg), p.ctypes.data_as(C.c_void_p))
    pool = np.empty(int(bound) * int(n_reads) + 64, dtype=np.uint8)
    tasks = np.empty(2 * int(n_reads) + 2, dtype=EXT_TASK)
    tread = np.empty(2 * int(n_reads) + 2, dtype=np.uint32)
    used = C.c_size_t(0)
    nt = L.bmh_taskgen_ext(C.byref(cfg), p.ctypes.data_as(C.c_void_p), C.c_int64(n_reads),
                           pool.ctypes.data_as(C.c_void_p), C.c_size_t(pool.nbytes), C.byref(used),
                           tasks.ctypes.data_as(C.c_void_p), C.c_int64(len(tasks)), tread.ctypes.data_as(C.c_void_p))
    if nt < 0:
        raise RuntimeError("taskgen capacity too small")
    pool = pool[: used.value + 16]
    pool[used.value:] = 0
    return pool, tasks[:nt].copy(), tread[:nt].copy()


def generate_seeds(params, n_reads, workload="150bp", seed=7):
    """Returns (pool uint8[], tasks SEED_TASK[]): one fused per-seed record per simulated (seeded) read."""
    from . import SEED_TASK, PARAMS
    L = _load()
    L.bmh_taskgen_seed.restype = C.c_int64
    cfg = Cfg(seed=seed, **WORKLOADS[workload])
    p = np.ascontiguousarray(np.asarray(params, dtype=PARAMS).reshape(()))
    bound = L.bmh_taskgen_pool_bound(C.byref(cfg), p.ctypes.data_as(C.c_void_p))
    pool = np.empty(int(bound) * int(n_reads) + 64, dtype=np.uint8)
    tasks = np.empty(int(n_reads) + 1, dtype=SEED_TASK)
    used = C.c_size_t(0)
    nt = L.bmh_taskgen_seed(C.byref(cfg), p.ctypes.data_as(C.c_void_p), C.c_int64(n_reads),
                            pool.ctypes.data_as(C.c_void_p), C.c_size_t(pool.nbytes), C.byref(used),
                            tasks.ctypes.data_as(C.c_void_p), C.c_int64(len(tasks)))
    if nt < 0:
        raise RuntimeError("taskgen capacity too small")
    pool = pool[: used.value + 16]
    pool[used.value:] = 0
    return pool, tasks[:nt].copy()


def split_reads_windows(pool, seeds):
    """(pool', seeds', reads_bytes): the same fused records over a pool laid out [all reads | all windows] -- the windows stand for
    reference bases resident in HBM, only pool'[:reads_bytes] travels per batch in a host-fed measurement."""
    from . import SEED_TASK
    L = _load()
    L.bmh_taskgen_split.restype = C.c_size_t
    t = np.ascontiguousarray(seeds, dtype=SEED_TASK).copy()
    out = np.empty(len(pool) + 128, dtype=np.uint8)
    rb = C.c_size_t(0)
    used = L.bmh_taskgen_split(pool.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p), C.c_int64(len(t)),
                               out.ctypes.data_as(C.c_void_p), C.c_size_t(out.nbytes), C.byref(rb))
    if not used:
        raise RuntimeError("split capacity too small")
    return out[:used], t, int(rb.value)


def split_queries_targets(gpool, gtasks):
    """(pool', tasks', query_bytes): the same ksw_global2 tasks over a pool laid out [all queries | all targets] -- the targets stand for
    reference windows resident in HBM (the region record fetches them there), only pool'[:query_bytes] travels per batch when host-fed."""
    from . import GLB_TASK
    L = _load()
    L.bmh_taskgen_split_glb.restype = C.c_size_t
    t = np.ascontiguousarray(gtasks, dtype=GLB_TASK).copy()
    out = np.empty(len(gpool) + 256, dtype=np.uint8)
    qb = C.c_size_t(0)
    used = L.bmh_taskgen_split_glb(gpool.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p), C.c_int64(len(t)),
                                   out.ctypes.data_as(C.c_void_p), C.c_size_t(out.nbytes), C.byref(qb))
    if not used:
        raise RuntimeError("split capacity too small")
    return out[:used], t, int(qb.value)


def generate_global(n_reads, workload="150bp", seed=11, wspread=32):
    """Returns (pool, tasks GLB_TASK[], cigar_words) -- one banded global alignment per simulated read."""
    from . import GLB_TASK
    L = _load()
    cfg = Cfg(seed=seed, **WORKLOADS[workload])
    pool = np.empty(int(n_reads) * (cfg.len_max * 3 + 64) + 64, dtype=np.uint8)
    tasks = np.zeros(int(n_reads) + 1, dtype=GLB_TASK)
    used, cw = C.c_size_t(0), C.c_uint64(0)
    nt = L.bmh_taskgen_glb(C.byref(cfg), C.c_int64(n_reads), C.c_int(wspread), pool.ctypes.data_as(C.c_void_p),
                           C.c_size_t(pool.nbytes), C.byref(used), tasks.ctypes.data_as(C.c_void_p),
                           C.c_int64(len(tasks)), C.byref(cw))
    if nt < 0:
        raise RuntimeError("taskgen capacity too small")
    pool = pool[: used.value + 16]
    pool[used.value:] = 0
    return pool, tasks[:nt].copy(), int(cw.value)


def generate_sw(params, n_mates, workload="150bp", seed=13, window=(300, 700), p_hit=0.7):
    """Returns (pool, tasks SW_TASK[]) -- one mate-rescue Smith-Waterman per simulated mate (mem_matesw's call of
    ksw_align2, reference bwamem_pair.c:147-148): the mate against an insert-size window of the reference."""
    from . import SW_TASK, PARAMS
    L = _load()
    L.bmh_taskgen_sw.restype = C.c_int64
    cfg = Cfg(seed=seed, **WORKLOADS[workload])
    p = np.ascontiguousarray(params, dtype=PARAMS)
    pool = np.empty(int(n_mates) * (cfg.len_max * 2 + window[1] + 64) + 64, dtype=np.uint8)
    tasks = np.zeros(int(n_mates) + 1, dtype=SW_TASK)
    used = C.c_size_t(0)
    nt = L.bmh_taskgen_sw(C.byref(cfg), p.ctypes.data_as(C.c_void_p), C.c_int64(n_mates), C.c_int(window[0]),
                          C.c_int(window[1]), C.c_double(p_hit), pool.ctypes.data_as(C.c_void_p),
                          C.c_size_t(pool.nbytes), C.byref(used), tasks.ctypes.data_as(C.c_void_p), C.c_int64(len(tasks)))
    if nt < 0:
        raise RuntimeError("taskgen capacity too small")
    pool = pool[: used.value + 16]
    pool[used.value:] = 0
    return pool, tasks[:nt].copy()
