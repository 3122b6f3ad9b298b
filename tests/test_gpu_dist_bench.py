"""The N > 1 launch path of bench.py on the 1-GPU box: `python -m torch.distributed.run --nproc-per-node 1 bench.py
--gpus 1 --force-dist` brings up the RCCL process group (backend "nccl"), takes the per-step all-gather, the barriers
and the MAX all-reduce of the timing exactly as an 8-rank run does (trainer/trainer.py:833 gather).
A scaling curve needs the driver's 8-GPU node; this pins that the distributed code path itself runs on RCCL."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_under_torchrun_world1_rccl():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist",
           "--steps", "1", "--warmup", "1", "--coalesce", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 1 and res["steps"] == 1 and res["value"] > 0
    assert res["config"]["parallelism"] == "dp1" and res["config"]["distributed"] == "nccl(RCCL) process group"
    assert 0 < res["roofline"]["frac"] < 1


def test_bench_self_launch_as_one_command_rccl():
    """`python bench.py --gpus N` as ONE command: no torchrun around it, no WORLD_SIZE - bench.py starts the ranks
    itself (N = 1 here through --self-launch; N > 1 takes the same branch on its own) and relays rank 0's line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--self-launch", "--steps", "1", "--warmup", "1",
           "--coalesce", "1", "--no-cpu-baseline", "--no-single-batch"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 1 and res["value"] > 0
    assert res["config"]["distributed"] == "nccl(RCCL) process group"
