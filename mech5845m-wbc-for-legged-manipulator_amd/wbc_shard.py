"""Batch sharding across the GPUs of one node: contiguous instance ranges, one process per GPU,
no data-path collective. The only communication is the benchmark's barrier and the max-over-ranks of the elapsed time."""


def shard_range(n_global, rank, world):
    """[lo, hi) of the instances owned by `rank`: contiguous, sizes differ by at most one, union = [0, n_global)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d of %d" % (rank, world))
    base, rem = divmod(n_global, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(value, dist=None, device=None):
    """max of a python float over all ranks (identity when torch.distributed is not initialised)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
