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


def record_bytes_of_ranks(total, world, bytes_per_filter):
    """bytes of packed records each rank holds when `total` filters are cut with shard_range: whole 64-filter tiles"""
    out = []
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        out.append((hi - lo + TILE - 1) // TILE * TILE * bytes_per_filter)
    return out


def native_comm_init(flt, rank, world, dist=None, device="cpu"):
    """Creates the library's own RCCL communicator for the BatchedFilter `flt` (fbus_ekf_comm_init): rank 0 draws the
    128-byte unique id, the others receive it over the control process group (a broadcast of 128 bytes -- the id is the only thing
    that ever travels outside the library's own collective)."""
    import torch
    raw, ok, why = bytes(128), 1, ""
    if rank == 0:
        try:
            raw = flt.comm_unique_id()
        except Exception as e:                      # no RCCL here: the other ranks must still get their broadcast
            ok, why = 0, str(e)
    if dist is not None and world > 1:
        t = torch.tensor([ok] + list(raw), dtype=torch.uint8, device=device)
        dist.broadcast(t, src=0)
        vals = t.cpu().tolist()
        ok, raw = vals[0], bytes(vals[1:])
    if not ok:
        raise RuntimeError("rank 0 could not create an RCCL unique id" + (f": {why}" if why else ""))
    flt.comm_init(raw, rank, world)


def gather_records_native(flt, local, bytes_of_rank=None, world=1):
    """the end-of-run gather through the LIBRARY (fbus_ekf_gather: RCCL on the handle's stream): returns the per-rank views of one
    device tensor holding every rank's records"""
    import torch
    sizes = list(bytes_of_rank) if bytes_of_rank is not None else [local.numel()] * world
    out = torch.empty(sum(sizes), dtype=torch.uint8, device=local.device)
    flt.gather(out, None if bytes_of_rank is None else sizes)
    flt.sync()
    views, off = [], 0
    for n in sizes:
        views.append(out[off:off + n])
        off += n
    return views
