"""Frame-batch sharding across the GPUs of one node: every FEC frame / SIMD batch is independent
given the mode tables, so ranks take disjoint, contiguous, batch-aligned ranges and never exchange data on the data
path. The only collectives are the bench's barrier and the max-over-ranks of the timed region."""


def shard_frames(total_frames, world_size, rank, align=32):
    """Contiguous range [lo, hi) of `total_frames` for `rank`; boundaries are multiples of `align` (a reference SIMD
    batch never straddles two GPUs). Ranges of all ranks are disjoint and cover [0, total_frames)."""
    assert 0 <= rank < world_size and total_frames >= 0 and align >= 1
    units = (total_frames + align - 1) // align
    base, rem = divmod(units, world_size)
    lo_u = rank * base + min(rank, rem)
    hi_u = lo_u + base + (1 if rank < rem else 0)
    return min(lo_u * align, total_frames), min(hi_u * align, total_frames)


def aggregate_timing(local_seconds, local_units, dist=None, device=None):
    """(max seconds over ranks, sum of units over ranks). With `dist` None (single process) returns the inputs."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_seconds, local_units
    import torch
    t = torch.tensor([local_seconds], dtype=torch.float64, device=device)
    u = torch.tensor([float(local_units)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())
