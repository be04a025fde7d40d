"""Data-parallel execution of the forward path: one process per GPU, batch shards, no data-path collective.

The reference wraps the model in single-process `torch.nn.DataParallel` (train.py:81, demo_RGBD.py:49): scatter the batch
along dim 0, replicate weights every iteration, gather outputs on GPU 0.  In eval every sample is independent (BatchNorm
uses running statistics), so the MI355X-native form is: weights resident per rank, each rank takes a
contiguous shard of the batch, and the only communication is an optional all_gather of the (small) outputs over
RCCL/xGMI — B x 1.03 MB per image.  `torch.distributed` backend "nccl" is RCCL on ROCm; tests use "gloo" on CPU.
"""
import torch


def shard_bounds(n, rank, world):
    """Contiguous, balanced [lo, hi) of n items for `rank` (first n % world ranks get one extra item)."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_batch(batch, rank, world):
    """batch: dict name -> tensor with batch dim 0.  Returns this rank's shard (views, no copy)."""
    n = next(iter(batch.values())).shape[0]
    lo, hi = shard_bounds(n, rank, world)
    return {k: v[lo:hi] for k, v in batch.items()}


def gather_outputs(tensors, n_total, dist_mod=None, group=None):
    """all_gather a list of per-rank output tensors (batch dim 0, possibly ragged shards) back into full-batch tensors in
    rank order.  With dist_mod None (single process) returns the inputs."""
    if dist_mod is None or not dist_mod.is_initialized() or dist_mod.get_world_size(group) == 1:
        return list(tensors)
    world = dist_mod.get_world_size(group)
    out = []
    q, r = divmod(n_total, world)
    maxn = q + (1 if r else 0)
    for t in tensors:
        pad = torch.zeros((maxn,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist_mod.all_gather(bufs, pad, group=group)
        parts = []
        for rk in range(world):
            lo, hi = shard_bounds(n_total, rk, world)
            parts.append(bufs[rk][: hi - lo])
        out.append(torch.cat(parts, 0))
    return out


def max_over_ranks(value, device, dist_mod=None):
    """max of a python float over all ranks (bench timing contract)."""
    if dist_mod is None or not dist_mod.is_initialized() or dist_mod.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist_mod.all_reduce(t, op=dist_mod.ReduceOp.MAX)
    return float(t.item())
