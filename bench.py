#!/usr/bin/env python
"""Headline benchmark: images/sec of CycleDiffusion on Stable-Diffusion-v1.4-shaped networks at
512x512, 99-step DPM-Encoder inversion + 99-step coupled decode with decoder CFG 3 (BASELINE.json
metric; SURVEY.md §8d "C2 headline"), synthetic weights / images / contexts (no checkpoints here).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL for the output gather)

A "step" = one pass of the hot path over one batch of 4 image triplets per GPU: the model API
forward = wrapper.encode (VAE encode + DPM-Encoder) + wrapper.forward (coupled decode + VAE decode),
i.e. exactly what Trainer.prediction_step times in the reference (trainer/trainer.py:788-789), followed
by the per-step output gather (trainer.py:833). Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F_UNET, F_VAE_ENC, F_VAE_DEC = 803.3e9, 1116.7e9, 2514.5e9   # FLOPs / sample
N_STEPS = 99
F_IMG = F_VAE_ENC + N_STEPS * F_UNET + N_STEPS * 2 * F_UNET + F_VAE_DEC   # 242.2 TFLOP / image
PEAK_TFLOPS = 2500.0  # dense 16-bit MFMA peak (MI355X_MICROARCH.md)


def host_cores():
    """CPU cores this process may actually use: min(affinity mask, cgroup-v2 quota). (The GPU box shows
    256 logical CPUs but caps the container at 16; oversubscribing a quota throttles everything.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(eng, un, vn, seed=0):
    """The CPU oracle (torch fp32 restatement of the reference path, oracle/) timed on this box's host
    cores on a bounded sample: one SD U-Net forward at batch 1 and one at batch 2 (the CFG pair), one
    VAE encode and one decode at 512x512; extrapolated linearly to 99 + 99 steps (all steps cost the
    same)."""
    from oracle import nets
    cores = host_cores()
    torch.set_num_threads(cores)
    ucfg = nets.OpenAIUNetCfg(in_channels=4, out_channels=4, model_channels=320, num_res_blocks=2,
                              channel_mult=(1, 2, 4, 4), attn_ds=(4, 2, 1), num_heads=8,
                              use_spatial_transformer=True, context_dim=768)
    vcfg = nets.VAECfg(ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2)
    usd = nets.synth_state_dict(eng.net_params(un), seed)
    vsd = nets.synth_state_dict(eng.net_params(vn), seed + 1)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        x = torch.randn(2, 4, 64, 64, generator=g)
        ctx = torch.randn(2, 77, 768, generator=g)
        t = torch.tensor([501, 501])
        nets.openai_unet(usd, ucfg, x[:1], t[:1], ctx[:1])  # warm-up (thread pool, allocator)
        t0 = time.time(); nets.openai_unet(usd, ucfg, x[:1], t[:1], ctx[:1]); t_u1 = time.time() - t0
        t0 = time.time(); nets.openai_unet(usd, ucfg, x, t, ctx); t_u2 = time.time() - t0
        img = torch.rand(1, 3, 512, 512, generator=g) * 2 - 1
        t0 = time.time(); mom = nets.vae_encode_moments(vsd, vcfg, img); t_e = time.time() - t0
        t0 = time.time(); nets.vae_decode(vsd, vcfg, mom[:, :4]); t_d = time.time() - t0
    per_img = t_e + N_STEPS * t_u1 + N_STEPS * t_u2 + t_d
    return {"value": 1.0 / per_img, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "oracle (torch fp32 CPU): 1 U-Net fwd @B=1 %.2fs + 1 @B=2 %.2fs + VAE enc %.2fs + dec %.2fs at "
                      "512x512, extrapolated linearly to 99 encode + 99 CFG decode steps (%.0f s/image)"
                      % (t_u1, t_u2, t_e, t_d, per_img)}


def pmc_traffic_per_launch():
    """HBM-side bytes per k_conv_gemm launch from the committed rocprofv3 PMC passes (profiles/README.md):
    FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE over one U-Net forward at B'=4 (encode) and B'=8 (CFG
    decode); a C2 step launches both forward types equally often. None when the summaries are absent."""
    vals = []
    for b in (4, 8):
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles",
                            "r1_conv_gemm_traffic_unet_b%d.json" % b)
        try:
            with open(path) as fh:
                vals.append(float(json.load(fh)["bytes_per_launch"]))
        except (OSError, KeyError, ValueError):
            return None
    return sum(vals) / len(vals)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4, help="image triplets per GPU per step (README.md:153)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--in-flight", type=int, default=4,
                    help="batches in flight per GPU: independent engine replicas (own HIP stream, workspace and "
                         "weights) driven by host threads; every step is still one batch of --batch triplets")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group even at world size 1 (exercises the gather path on a 1-GPU box)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # host-side weight synthesis / oracle: stay inside the CPU quota, shared by the ranks of the node
    torch.set_num_threads(max(1, host_cores() // max(1, world)))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 or world > 1 or a.force_dist:
        assert world == a.gpus, "launch with torch.distributed.run --nproc-per-node %d" % a.gpus
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local)

    import cycle_diffusion_amd as cda  # noqa: F401  (builds nothing: the .so travels in-tree)
    from cycle_diffusion_amd.utils.config_utils import get_config
    from cycle_diffusion_amd.utils.program_utils import get_model
    from cycle_diffusion_amd.parallel import gather_outputs, run_in_flight, shard_range

    os.environ["LOCAL_RANK"] = str(local)
    os.environ["CYCLEDIFF_SYNTHETIC_WEIGHTS"] = "1"  # no checkpoints in this tree: seeded synthetic weights (opt-in)
    args = get_config("experiments/bench_sd_c2.cfg", config_root=os.path.join(ROOT, "config"))
    # A single stream of these kernels leaves the GPU under-occupied (most launches are wait-bound at 1-3
    # workgroups per CU): two independent batches in flight on two HIP streams raise whole-GPU
    # throughput ~1.4x. Replica r owns stream r, its own engine (workspace, split-K scratch) and weights.
    n_rep = max(1, a.in_flight)
    os.environ["CYCLEDIFF_SHARE_SYNTH"] = "1" if n_rep > 1 else "0"  # generate the synthetic weights once per rank
    replicas = []
    for r in range(n_rep):
        st = torch.cuda.Stream(device=dev) if n_rep > 1 else torch.cuda.current_stream(dev)
        with torch.cuda.stream(st):
            torch.manual_seed(0)  # same weights on every rank and replica (main.py:66)
            replicas.append((st, get_model(args.model.name)(args).eval()))
    model = replicas[0][1]
    eng = model.gan_wrapper.engine
    cda.Engine._SYNTH_CACHE.clear()

    # synthetic batch: global batch = B * world, rank r takes its contiguous slice (ShardSampler, trainer.py:288-293)
    B = a.batch
    lo, hi = shard_range(B * world, world, rank)
    g = torch.Generator().manual_seed(1)
    images = torch.rand(B * world, 3, 512, 512, generator=g)[lo:hi].to(dev)
    sample_id = torch.arange(lo, hi, device=dev)
    src = ["source prompt %d" % i for i in range(lo, hi)]
    tgt = ["target prompt %d" % i for i in range(lo, hi)]
    torch.manual_seed(4 + rank)  # per-rank noise streams

    def compute(r):
        st, m = replicas[r]
        torch.cuda.set_device(dev)  # the current device is per host thread
        with torch.cuda.stream(st), torch.no_grad():
            return m(sample_id=sample_id, original_image=images, encode_text=src, decode_text=tgt)

    def gather(r, res):
        (orig, img), loss, _ = res
        torch.cuda.current_stream(dev).wait_stream(replicas[r][0])
        return gather_outputs((orig, img), loss)  # one all-gather per eval step (trainer.py:833)

    def step(r=0):
        return gather(r, compute(r))

    def run_steps(n):
        """n steps, up to n_rep of them in flight: one host thread per replica computes, then the main thread does
        that round's all-gathers in step order (parallel.run_in_flight)."""
        return run_in_flight(n, n_rep, compute, gather)

    def sync():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(a.warmup):  # replica by replica: the first one tunes the GEMM table, the others reuse it
        for r in range(n_rep):
            step(r)
            torch.cuda.synchronize(dev)
    sync()
    t0 = time.perf_counter()
    out = run_steps(a.steps)
    sync()
    dt = time.perf_counter() - t0
    if dist.is_initialized():
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(out[0][1]).all()

    res = None
    # roofline of the dominant kernel (implicit-GEMM conv / GEMM family): one more identical step (all
    # ranks take part in its gather) with per-launch HIP events on rank 0's engine stream;
    # achieved = sum(2*M*N*K) / sum(launch durations)
    if rank == 0:
        eng.prof_enable(True)
    step()
    sync()
    if rank == 0:
        ips = a.steps * B * world / dt
        n_launch, k_ms, k_flops = eng.prof_collect()
        eng.prof_enable(False)
        ach = k_flops / (k_ms * 1e-3) / 1e12
        traffic = pmc_traffic_per_launch()
        res = {
            "metric": "images/sec, SD-v1.4 512px CycleDiffusion 100+100 steps, 1/2/4/8 MI355X", "value": ips,
            "unit": "images/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp16" if eng.lib.cd_act_format() == 1 else "bf16", "data": "synthetic",
            "config": {"workload": "C2: Stable-Diffusion-v1.4-shaped U-Net + KL-f8 VAE, 512x512, custom_steps=99 "
                                   "white_box_steps=100 eta=0.1 skip 0, 1 trial, encoder scale 1, decoder CFG 3",
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": "dp%d" % world,
                       "batches_in_flight_per_gpu": n_rep,
                       "weights": model.gan_wrapper.weights_origin, "flop_per_image": F_IMG},
            "roofline": {"bound": "mfma", "kernel": "k_conv_gemm (all tile instantiations)",
                         "achieved": ach, "peak": PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_TFLOPS,
                         "traffic": traffic, "traffic_unit": "HBM-side bytes per launch (PMC, profiles/)",
                         "launches_per_step": n_launch, "kernel_ms_per_step": k_ms,
                         "algorithmic_tflop_per_step": k_flops / 1e12,
                         "whole_path_frac": ips * F_IMG / 1e12 / (world * PEAK_TFLOPS)},
        }
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(eng, model.gan_wrapper.unet, model.gan_wrapper.vae)
        print(json.dumps(res), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return res


if __name__ == "__main__":
    main()
