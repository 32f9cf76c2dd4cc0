"""Shard arithmetic and the end-of-run gather for multi-GPU runs (one process per GPU).

Filters are independent (no cross-filter term in ImuUpdate.m / MeasureUpdate.m), so the
batch is cut into contiguous, 64-aligned ranges -- one per rank -- and nothing is
exchanged while stepping.  The only collective is one gather of the packed records at
the end (RCCL over xGMI when the tensors are on GPUs, gloo in the CPU tests).
"""
TILE = 64


def shard_range(total, rank, world):
    """[lo, hi) of the filters rank owns: contiguous, tile (64) aligned, sizes differ by <= one tile."""
    tiles = (total + TILE - 1) // TILE
    lo_t = tiles * rank // world
    hi_t = tiles * (rank + 1) // world
    return min(lo_t * TILE, total), min(hi_t * TILE, total)


def weak_range(per_rank, rank):
    """[lo, hi) in the global filter index space when every rank owns per_rank filters (weak scaling)."""
    return per_rank * rank, per_rank * (rank + 1)


def gather_records(local, dist=None, world=1):
    """All-gathers equally sized 1-D record tensors; returns the list of per-rank tensors
    (a single-element list without a process group)."""
    if dist is None:
        return [local]
    import torch
    out = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(out, local)
    return out


def gather_records_ragged(local, dist=None, world=1, device="cpu"):
    """All-gather of 1-D record tensors whose lengths may differ by a tile (strong scaling with a total that is not
    a multiple of 64 x world): padded to the longest shard for the collective, trimmed afterwards."""
    if dist is None:
        return [local]
    import torch
    n = torch.tensor([local.numel()], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(x.item()) for x in sizes]
    longest = max(sizes)
    padded = local
    if local.numel() < longest:
        padded = torch.zeros(longest, dtype=local.dtype, device=local.device)
        padded[:local.numel()] = local
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded)
    return [o[:k] for o, k in zip(out, sizes)]


def sum_over_ranks(value, dist=None, world=1, device="cpu"):
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def max_over_ranks(value, dist=None, world=1, device="cpu"):
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
