"""CPU, world_size 2 over gloo: the N>1 path of the benchmark / model API — contiguous sharding of
the triplets and the single per-step all-gather — reproduces the single-process batch order
(trainer/trainer.py:43-61, 288-293)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cycle_diffusion_amd.parallel import gather_outputs, shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = torch.arange(n_items * 6, dtype=torch.float32).view(n_items, 2, 3)
    lo, hi = shard_range(n_items, world, rank)
    local_in = full[lo:hi]
    local_out = local_in * 2 + 1                       # stand-in for the per-rank translation
    (orig, img), loss = gather_outputs((local_in, local_out), torch.zeros(hi - lo))
    ok = torch.equal(orig, full) and torch.equal(img, full * 2 + 1) and loss.shape[0] == n_items
    q.put((rank, bool(ok), lo, hi))
    dist.barrier()
    dist.destroy_process_group()


def _worker_ragged(rank, world, port, n_items, bsz, q):
    """n_items % (bsz * world) != 0: every rank still runs the same number of full batches (wrap-around padding),
    the gathers never see ragged shapes, and truncation to num_total_examples restores the dataset."""
    from cycle_diffusion_amd.parallel import global_order, shard_indices
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = torch.arange(n_items * 3, dtype=torch.float32).view(n_items, 3)
    steps = shard_indices(n_items, bsz, world, rank)
    rows, ids = [], []
    for idx in steps:
        assert len(idx) == bsz
        local = full[idx]
        (img, none_leaf, d), sid = gather_outputs((local * 2, None, {"k": local + 1}), torch.tensor(idx))
        assert none_leaf is None and img.shape[0] == bsz * world and torch.equal(d["k"], img / 2 + 1)
        rows.append(img)
        ids.append(sid)
    rows, ids = torch.cat(rows), torch.cat(ids)
    order = global_order(n_items, bsz, world)
    ok = ids.tolist() == order and torch.equal(rows, full[order] * 2)
    # a scalar leaf becomes 1-d, and truncation drops the padding of a single gather
    (sc,), _ = gather_outputs((torch.tensor(float(rank)),), None, num_total_examples=world - 1 if world > 1 else 1)
    ok = ok and sc.tolist() == [float(r) for r in range(max(1, world - 1))]
    # dataset order is recovered by sample id (what main.py does)
    seen = {}
    for i, r in zip(ids.tolist(), rows):
        seen[i] = r
    ok = ok and sorted(seen) == list(range(n_items)) and all(torch.equal(seen[i], full[i] * 2) for i in seen)
    q.put((rank, bool(ok), len(steps)))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_sampler_semantics_single_process():
    from cycle_diffusion_amd.parallel import global_order, shard_indices
    # HF ShardSampler docstring example: 2 processes, batch 4, 16 items
    assert shard_indices(16, 4, 2, 0) == [[0, 1, 2, 3], [8, 9, 10, 11]]
    assert shard_indices(16, 4, 2, 1) == [[4, 5, 6, 7], [12, 13, 14, 15]]
    # 10 items: padded by wrap-around to 16
    assert shard_indices(10, 4, 2, 0) == [[0, 1, 2, 3], [8, 9, 0, 1]]
    assert shard_indices(10, 4, 2, 1) == [[4, 5, 6, 7], [2, 3, 4, 5]]
    assert global_order(10, 4, 2)[:10] == list(range(10))
    assert shard_indices(0, 4, 2, 0) == [] and shard_indices(3, 4, 1, 0) == [[0, 1, 2, 0]]


def test_ragged_dataset_gather_world2():
    world, n_items, bsz = 2, 10, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ragged, args=(r, world, port, n_items, bsz, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert all(n == 2 for _, _, n in res)


def test_shard_ranges_partition_the_batch():
    for n, w in ((8, 2), (64, 8), (32, 8), (4, 1)):
        spans = [shard_range(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))


def test_gather_reproduces_global_order_world2():
    world, n_items = 2, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res), res
    assert sorted((lo, hi) for _, _, lo, hi in res) == [(0, 4), (4, 8)]


def _worker_in_flight(rank, world, port, q):
    """bench.py's N>1 schedule on CPU: 3 replicas in flight, 7 steps, jittered compute threads, one all-gather per step
    issued by the main thread in step order on every rank."""
    import random
    import time
    from cycle_diffusion_amd.parallel import run_in_flight
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = random.Random(100 + rank)
    counter = {"step": 0}
    seen = []

    def compute(r):
        time.sleep(rng.random() * 0.05)  # threads finish in a different order on every rank
        return torch.full((2, 3), float(rank * 100 + r))

    def finish(r, res):
        out, _ = gather_outputs((res,), None)
        seen.append((counter["step"], r, out[0][:, 0].tolist()))
        counter["step"] += 1
        return out[0]

    last = run_in_flight(7, 3, compute, finish)
    ok = len(seen) == 7 and [s for s, _, _ in seen] == list(range(7))
    ok = ok and [r for _, r, _ in seen] == [0, 1, 2, 0, 1, 2, 0]  # rounds of 3, 3, 1
    ok = ok and all(v == [0.0 + r, 0.0 + r, 100.0 + r, 100.0 + r] for _, r, v in seen)  # rank-major, same replica
    ok = ok and last.shape == (4, 3)
    try:
        run_in_flight(2, 2, lambda r: (_ for _ in ()).throw(ValueError("boom")) if r == 1 else 0, lambda r, x: x)
        ok = False
    except ValueError:
        pass
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_in_flight_rounds_keep_collectives_in_step_order_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_in_flight, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def test_shard_sampler_at_world_8_with_wrap_around_padding():
    """ShardSampler semantics (trainer/trainer.py:288-293) at the rank counts BASELINE configs 4 / 5 name: 8 ranks x batch 8
    (C4) and 8 x 4 (C5) on datasets that do not fill the last global batch - every rank gets the same number of full batches,
    the padding wraps around to the first samples, every real sample appears exactly once among the non-padding slots."""
    from cycle_diffusion_amd.parallel import global_order, shard_indices, shard_padding
    for n, bs in ((150, 8), (100, 4), (64, 8), (5, 4), (33, 4)):
        world, gb = 8, bs * 8
        total = -(-n // gb) * gb
        per_rank = [shard_indices(n, bs, world, r) for r in range(world)]
        pads = [shard_padding(n, bs, world, r) for r in range(world)]
        assert len({len(p) for p in per_rank}) == 1 and all(len(b) == bs for p in per_rank for b in p)
        order = global_order(n, bs, world)
        assert len(order) == total and order[:n] == list(range(n))
        assert order[n:] == [i % n for i in range(total - n)]  # wrap-around, as `indices += indices[:pad]` repeated
        real = [i for r in range(world) for b, pd in zip(per_rank[r], pads[r]) for i, is_pad in zip(b, pd) if not is_pad]
        assert sorted(real) == list(range(n))
        assert sum(is_pad for r in range(world) for pd in pads[r] for is_pad in pd) == total - n
