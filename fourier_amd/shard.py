"""Multi-GPU host logic: the batch index shards contiguously across ranks, no data-path collective.

SURVEY.md section 8(e): transforms are independent (the reference's `Fft::transform` takes one slice,
fourier-algorithms/src/fft.rs:51-61), so rank g of G owns transforms [floor(g*B/G), floor((g+1)*B/G)).
Only timings / checksums are reduced across ranks (torch.distributed; RCCL on GPUs, gloo in CPU tests).
"""


def batch_shard(global_batch, world, rank):
    """Contiguous [begin, end) range of transform indices owned by `rank`."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return (global_batch * rank) // world, (global_batch * (rank + 1)) // world


def owner_of(b, global_batch, world):
    """Rank owning transform b (inverse of batch_shard)."""
    for r in range(world):
        lo, hi = batch_shard(global_batch, world, r)
        if lo <= b < hi:
            return r
    raise ValueError("transform index out of range")


def reduce_max_seconds(seconds, dist=None, device=None):
    """Whole-job time = max over ranks (each rank barrier+sync brackets its own timed region)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    import torch

    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_rows(local_rows, dist=None):
    """Gather per-rank numpy row blocks (used for checksums / parity samples, not in the data path)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [local_rows]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, local_rows)
    return out
