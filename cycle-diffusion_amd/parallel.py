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


def run_in_flight(n_steps, n_replicas, compute, finish):
    """Run `n_steps` independent steps with up to `n_replicas` of them in flight (bench.py: one engine replica per
    HIP stream). Each round starts one host thread per replica running `compute(replica) -> result`; when the round's
    threads have joined, the MAIN thread calls `finish(replica, result)` for the round's steps in step order - that is
    where collectives go, so every rank issues them in the same order no matter how its threads were scheduled.
    Returns the last `finish` value. A replica that raises aborts the run with that exception."""
    import threading
    out, done = None, 0
    while done < n_steps:
        k = min(n_replicas, n_steps - done)
        res, err = {}, {}

        def work(r):
            try:
                res[r] = compute(r)
            except BaseException as e:  # noqa: BLE001 - re-raised on the main thread
                err[r] = e

        if k == 1:
            work(0)
        else:
            ths = [threading.Thread(target=work, args=(r,)) for r in range(k)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
        if err:
            raise err[min(err)]
        for r in range(k):
            out = finish(r, res[r])
        done += k
    return out
