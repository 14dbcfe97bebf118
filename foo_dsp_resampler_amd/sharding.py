"""Stream sharding across GPUs.  Streams (and channels) are independent (rate/rate_base.h:533-540: one
rate_t per channel, only read-only coefficient tables shared), so N GPUs = N disjoint shards and the data
path needs no collective.  torch.distributed is used only to agree on timings."""


def shard_range(total, world, rank):
    """Contiguous, balanced partition of `total` streams: returns (first, count) for `rank`."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, rem = divmod(total, world)
    return rank * base + min(rank, rem), base + (1 if rank < rem else 0)


def max_over_ranks(values, dist=None, device="cpu"):
    """Element-wise MAX of a list of floats over all ranks (identity without a process group)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(values)
    import torch
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t]


def job_throughput(units_per_rank, elapsed_max_s):
    """Whole-job rate: units all ranks processed / the slowest rank's time."""
    return sum(units_per_rank) / elapsed_max_s
