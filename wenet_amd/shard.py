"""Multi-GPU sharding of independent captures: unit = one capture, no exchange step.

Capture i goes to rank i % world_size; every rank runs the full chain on its own captures on its own
GPU.  No collective touches the data path -- the only communication is the bookkeeping the caller
wants (a barrier around timing, a reduction of packet counters)."""
from __future__ import annotations


def shard_indices(n_items: int, rank: int, world_size: int):
    """Indices of the captures rank `rank` owns (round-robin: capture index mod n_gpus)."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    return list(range(rank, n_items, world_size))


def gather_counts(local_counts, n_items: int, rank: int, world_size: int, dist=None):
    """Combine per-capture counters computed on each rank into one list indexed by capture
    (all_reduce SUM of a zero-padded vector; works with gloo on CPU and nccl/RCCL on GPUs)."""
    import torch

    idx = shard_indices(n_items, rank, world_size)
    assert len(idx) == len(local_counts)
    v = torch.zeros(n_items, dtype=torch.int64)
    for i, c in zip(idx, local_counts):
        v[i] = int(c)
    if dist is not None and world_size > 1:
        if dist.get_backend() == "nccl":
            v = v.cuda()
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        v = v.cpu()
    return v.tolist()
