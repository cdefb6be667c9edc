"""Read partitioning over the GPUs of one node: reads are independent, so rank r of W owns
a contiguous block of read indices, the index is replicated per GPU and the data path needs no
collective.  torch.distributed (backend nccl = RCCL on the GPU box, gloo in CPU tests) is only used
for the barrier and for reducing the timing / counts that bench.py reports."""


def shard_range(n_total, world, rank):
    """Contiguous block [lo, hi) of read indices owned by `rank` (blocks differ by at most one read)."""
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def weak_shard_first_index(reads_per_rank, rank):
    """Weak scaling: every rank generates / aligns its own reads_per_rank reads."""
    return int(reads_per_rank) * int(rank)


def reduce_timing_and_counts(dist, device, seconds, counts):
    """MAX over ranks of the timed region, SUM over ranks of the unit counts. dist may be None (1 rank)."""
    import torch
    if dist is None:
        return float(seconds), [float(c) for c in counts]
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([float(x) for x in counts], dtype=torch.float64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), [float(x) for x in c.tolist()]
