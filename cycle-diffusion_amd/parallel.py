"""Data-parallel plumbing: one process per GPU, triplets sharded contiguously, ONE collective per
eval step — the all-gather of the outputs (the reference's only data-path collective:
distributed_concat -> dist.all_gather, trainer/trainer.py:43-61, called at :833; sharding:
ShardSampler, trainer.py:288-293). backend "nccl" is RCCL over xGMI on ROCm; "gloo" in CPU tests.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, world_size, rank):
    """Contiguous slice of ONE global batch for `rank` when the batch divides evenly (bench.py: global batch =
    B_local * world). For a whole dataset use shard_indices (ShardSampler semantics incl. wrap-around padding)."""
    per = (n_items + world_size - 1) // world_size
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)


def shard_indices(n_items, batch_size, world_size, rank):
    """HF `ShardSampler(dataset, batch_size, num_processes, process_index)` as the reference's eval loader uses it
    (trainer/trainer.py:288-293): the index list is padded by WRAP-AROUND to a multiple of batch_size * world_size,
    cut in global batches of that size, and each global batch is split contiguously - rank r takes
    [r * batch_size, (r + 1) * batch_size). Every rank therefore gets the same number of full batches (collectives
    never see ragged shapes); the padding is dropped after the gather by gather_outputs(..., num_total_examples).
    Returns the list of per-step index lists for `rank`."""
    if n_items <= 0:
        return []
    gb = batch_size * world_size
    total = ((n_items + gb - 1) // gb) * gb
    idx = list(range(n_items))
    while len(idx) < total:
        idx += idx[: total - len(idx)]
    return [idx[g0 + rank * batch_size: g0 + (rank + 1) * batch_size] for g0 in range(0, total, gb)]


def shard_padding(n_items, batch_size, world_size, rank):
    """Same shape as shard_indices: True where a slot holds wrap-around padding (its position in the padded index list is
    >= n_items) - the rows the reference drops after the gather (`num_total_examples`); a driver that writes per-sample
    files skips them where they are produced."""
    if n_items <= 0:
        return []
    gb = batch_size * world_size
    total = ((n_items + gb - 1) // gb) * gb
    return [[(g0 + rank * batch_size + j) >= n_items for j in range(batch_size)] for g0 in range(0, total, gb)]


def global_order(n_items, batch_size, world_size):
    """dataset index of every row of the concatenated per-step gathers (step-major, then rank-major), padding
    included - what distributed_concat + nested_concat produce in the reference's eval loop (trainer.py:825-840)."""
    steps = [shard_indices(n_items, batch_size, world_size, r) for r in range(world_size)]
    out = []
    for s in range(len(steps[0]) if steps else 0):
        for r in range(world_size):
            out += steps[r][s]
    return out


def _concat(t, num_total_examples=None):
    """distributed_concat on one leaf (trainer/trainer.py:43-61): all_gather in rank order, 0-d tensors become 1-d,
    optional truncation of the sampler's padding."""
    if t is None:
        return None
    if not (dist.is_available() and dist.is_initialized()):
        out = t if t.dim() > 0 else t[None]
    else:
        t = t.contiguous()
        outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(outs, t)  # same shape on every rank: shard_indices pads the last global batch
        out = torch.cat([o if o.dim() > 0 else o[None] for o in outs], dim=0)
    return out[:num_total_examples] if num_total_examples is not None else out


def _gather_nested(t, num_total_examples=None):
    if isinstance(t, (tuple, list)):
        return type(t)(_gather_nested(x, num_total_examples) for x in t)
    if isinstance(t, dict):
        return type(t)({k: _gather_nested(v, num_total_examples) for k, v in t.items()})
    return _concat(t, num_total_examples)


def gather_outputs(tensors, loss=None, num_total_examples=None):
    """nested tuple / list / dict of per-rank tensors (None leaves pass through) -> same structure concatenated along
    dim 0 in rank order; `num_total_examples` truncates the wrap-around padding of the last global batch."""
    return _gather_nested(tensors, num_total_examples), _concat(loss, num_total_examples)


def run_in_flight(n_steps, n_replicas, compute, finish, pass_index=False):
    """Run `n_steps` independent steps with up to `n_replicas` of them in flight (bench.py: one engine replica per
    HIP stream). Each round starts one host thread per replica running `compute(replica) -> result`; when the round's
    threads have joined, the MAIN thread calls `finish(replica, result)` for the round's steps in step order - that is
    where collectives go, so every rank issues them in the same order no matter how its threads were scheduled.
    With pass_index the call is `compute(replica, step_index)`.
    Returns the last `finish` value. A replica that raises aborts the run with that exception."""
    import threading
    out, done = None, 0
    while done < n_steps:
        k = min(n_replicas, n_steps - done)
        res, err = {}, {}

        def work(r, base=done):
            try:
                res[r] = compute(r, base + r) if pass_index else compute(r)
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
