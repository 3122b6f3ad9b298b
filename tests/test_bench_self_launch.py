"""`python bench.py --gpus N` given as ONE plain command (the driver's invocation shape) must run: with N > 1 and no
WORLD_SIZE in the environment bench.py starts its N ranks itself under torch.distributed.run and rank 0 prints the one
JSON line (the reference's ranks come from `python -m torch.distributed.launch --nproc_per_node 8 main.py`,
README.md:153; trainer/trainer.py:174-179 reads the env they set). Driven here at world size 2 in --dry-run mode: gloo,
CPU tensors, a stand-in for the engine call - the host plumbing (sharding, one all-gather per step in step order,
barriers, MAX over ranks) is the same code the GPU run takes."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, drop=("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, r.stdout[-2000:]  # ONE line, from rank 0
    return json.loads(lines[0])


def test_plain_command_with_gpus_2_starts_its_own_ranks():
    res = _run(["--gpus", "2", "--steps", "5", "--warmup", "2", "--coalesce", "2", "--dry-run"])
    assert res["n_gpus"] == 2 and res["steps"] == 5 and res["warmup"] == 2 and res["dry_run"] is True
    assert res["config"]["parallelism"] == "dp2" and res["config"]["global_batch"] == 8
    assert res["config"]["distributed"] == "gloo process group"
    assert res["scaling"] == "weak" and res["value"] > 0 and res["ms_per_step"] > 0


def test_self_launch_at_one_rank_and_plain_single_process():
    res = _run(["--gpus", "1", "--self-launch", "--steps", "3", "--warmup", "0", "--dry-run"])
    assert res["n_gpus"] == 1 and res["config"]["distributed"] == "gloo process group"
    res = _run(["--steps", "3", "--warmup", "0", "--dry-run"])
    assert res["n_gpus"] == 1 and res["config"]["distributed"] == "single process"


def test_world_size_mismatch_is_an_error_not_a_hang():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "1"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "does not match --gpus" in r.stderr


def _cores():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def test_eight_ranks_dry_run_c4_and_c5_batches():
    """BASELINE configs 4 and 5 name 8 GPUs: SD-v1.4 batch 64 (8 per GPU) and AFHQ batch 32 (4 per GPU). The 8-GPU node is the
    driver's; here the same launch path runs 8 gloo ranks on the host: every rank's id arrives through one all-gather
    (`ranks_seen`), every step gathers one full global batch, and each rank caps its host threads at cores // 8."""
    res = _run(["--gpus", "8", "--steps", "3", "--warmup", "0", "--coalesce", "2", "--batch", "8", "--dry-run"])
    assert res["n_gpus"] == 8 and res["ranks_seen"] == list(range(8))
    assert res["config"]["parallelism"] == "dp8" and res["config"]["global_batch"] == 64 and res["config"]["batch_per_gpu"] == 8
    assert res["host_threads_per_rank"] == max(1, _cores() // 8)
    res = _run(["--gpus", "8", "--workload", "c5", "--steps", "2", "--warmup", "0", "--dry-run"])
    assert res["ranks_seen"] == list(range(8)) and res["config"]["global_batch"] == 32 and res["config"]["batch_per_gpu"] == 4
