"""Data-parallel plumbing: one process per GPU, triplets sharded contiguously, ONE collective per
eval step — the all-gather of the outputs (the reference's only data-path collective:
distributed_concat -> dist.all_gather, trainer/trainer.py:43-61, called at :833; sharding:
ShardSampler, trainer.py:288-293). backend "nccl" is RCCL over xGMI on ROCm; "gloo" in CPU tests.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, world_size, rank):
    """Contiguous slice of a global batch for `rank` (global chunks of B_local*world split contiguously)."""
    per = (n_items + world_size - 1) // world_size
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)


def _concat(t):
    if not (dist.is_available() and dist.is_initialized()):
        return t
    t = t.contiguous()
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return torch.cat(outs, dim=0)


def gather_outputs(tensors, loss=None):
    """nested tuple of per-rank tensors -> same structure concatenated along dim 0 in rank order."""
    if isinstance(tensors, (tuple, list)):
        out = type(tensors)(_gather_nested(t) for t in tensors)
    else:
        out = _concat(tensors)
    return out, (_concat(loss) if loss is not None else None)


def _gather_nested(t):
    if isinstance(t, (tuple, list)):
        return type(t)(_gather_nested(x) for x in t)
    return _concat(t)
