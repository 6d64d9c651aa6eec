"""Static sharding helpers shared by bench.py and the multi-process tests.

The hot path shards over independent units (extension tasks / reads): every rank owns its own
batch, there is NO collective on the data path.  torch.distributed is used only to line the
ranks up (barrier) and to combine the timing and the unit counts for the report."""
import torch
import torch.distributed as dist


def shard_seed(base_seed, rank):
    """Distinct, reproducible workload seed per rank (weak scaling: same size, different reads)."""
    return int(base_seed) + 1000 * int(rank)


def shard_ranges(n, world):
    """Contiguous split of n units over `world` shards -- same formula as bmh_extend_batch_sharded (api.hip)."""
    return [(n * g // world, n * (g + 1) // world) for g in range(world)]


def reduce_report(elapsed_s, n_reads, n_tasks, device=None):
    """max-over-ranks elapsed time, sum-over-ranks unit counts.  Works on NCCL(RCCL) and gloo."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(elapsed_s), float(n_reads), float(n_tasks)
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([n_reads, n_tasks], dtype=torch.float64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), float(c[0].item()), float(c[1].item())
