#!/usr/bin/env python
"""Headline benchmark: images/sec of CycleDiffusion on Stable-Diffusion-v1.4-shaped networks at
512x512, 99-step DPM-Encoder inversion + 99-step coupled decode with decoder CFG 3 (BASELINE.json
metric; SURVEY.md §8d "C2 headline"), synthetic weights / images / contexts (no checkpoints here).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: one rank per GPU, RCCL for the output gather. Either the caller starts the ranks - `python -m
  torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`,
  RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env - or the plain command is given and bench.py starts them
  itself: with --gpus N > 1 and no WORLD_SIZE in the environment it re-executes under torch.distributed.run on
  127.0.0.1 with a free port and relays rank 0's JSON line; the reference's ranks are started the same way,
  README.md:153 `python -m torch.distributed.launch --nproc_per_node 8 ... main.py`, trainer/trainer.py:174-179.)

A "step" = one pass of the hot path over one batch of 4 image triplets per GPU: the model API
forward = wrapper.encode (VAE encode + DPM-Encoder) + wrapper.forward (coupled decode + VAE decode),
i.e. exactly what Trainer.prediction_step times in the reference (trainer/trainer.py:788-789), followed
by the per-step output gather (trainer.py:833). Prints ONE JSON line on rank 0.

Steps are issued `--coalesce` at a time (default: 16 for C2, 4 otherwise; a remainder forms one smaller tail set,
`config.launch_sets`): the engine folds the queued batches into
ONE launch set (C2: 64 images through the DPM-Encoder, 128 rows through the CFG decode, the first stage in calls of 32
images), so every GEMM sees 16x the rows with one copy of the weights; each step still gets its own all-gather, in step
order. `--coalesce 8` / `4` / `1` are the rounds-2b..4 / round-2a / round-1 operating points (8 against 16 on one box:
3.35 -> 3.44 images/s, profiles/r5_bench_lines_coalesce_8_vs_16.json; every one of them is pinned to the reference by
tests/test_gpu_e2e_fullsize.py's folded-batch tests). The line also carries `single_batch` - launch sets of ONE step, the
operating point of the reference's own driver loop. `main.py --fold N` is the same look-ahead for real data. `--in-flight R`
additionally keeps R such launch sets running on R independent engines / HIP streams.

Other BASELINE.json configurations: `--workload c3` (LDM text2img-large shapes, 256 x 256, batch 16) and
`--workload c5r` (AFHQ improved-DDPM pair, 256 x 256, batch 4; REDUCED chain custom_steps 100 / es_steps 85 /
refine_steps 10 = the reference cfg's 1000 / 850 / 100 divided by 10 - labelled as such in the line); `--workload c5`
is the reference chain itself (1799 U-Net evaluations per image: use --steps 1..2). Both run the fp32 networks in the
split-fp16 mode (`--precision fp32x3`, DESIGN.md 5: same parity floors, 2.7x the fp32 path) unless `--precision fp32`
asks for the reference's own arithmetic; `--workload c2e` is the reference's SD ensemble experiment (540 candidates per
image, ~93 s per image: use --steps 1 --warmup 0).
"""
import argparse
import json
import os
import socket
import subprocess
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
PEAK_F32_TFLOPS = 157.3  # fp32 MFMA peak (v_mfma_f32_32x32x2_f32)

WORKLOADS = {
    "c2": dict(cfg="experiments/bench_sd_c2.cfg", res=512, batch=4, text=True, flop_per_image=F_IMG, coalesce=16,
               metric="images/sec, SD-v1.4 512px CycleDiffusion 100+100 steps, 1/2/4/8 MI355X",
               name="C2: Stable-Diffusion-v1.4-shaped U-Net + KL-f8 VAE, 512x512, custom_steps=99 "
                    "white_box_steps=100 eta=0.1 skip 0, 1 trial, encoder scale 1, decoder CFG 3"),
    # the reference's actual SD experiment: per image 90 DPM-Encoder runs (15 trials x 6 skips, scale 1: one forward per
    # step, ddim.py:550) and 540 decodes (6 scales; scale 1 decodes need one forward per step, the others two), then 540
    # VAE decodes and a directional-CLIP ranking. sum over skips of (99 - skip) = 414 steps
    "c2e": dict(cfg="experiments/translate_text2img256_stable_diffusion_stochastic_1.cfg", res=512, batch=1, text=True,
                coalesce=1,
                flop_per_image=F_VAE_ENC + 15 * 414 * F_UNET + 15 * 414 * 11 * F_UNET + 540 * F_VAE_DEC,
                metric="images/sec, SD-v1.4 512px CycleDiffusion ensemble as the reference runs it: 15 trials x 6 skips x 6 "
                       "decoder scales = 540 candidates per image, directional-CLIP ranked",
                name="C2-ensemble: translate_text2img256_stable_diffusion_stochastic_1.cfg as written (custom_steps=99 "
                     "white_box_steps=100 eta=0.1, n_trials=15, skip_steps [15..50], decoder scales [1..5]), 512x512"),
    "c3": dict(cfg="experiments/bench_ldm_c3.cfg", res=256, batch=16, text=True,
               flop_per_image=272.7e9 + 99 * 182.1e9 + 99 * 2 * 182.1e9 + 622.2e9,   # 55.0 TFLOP
               metric="images/sec, LDM text2img-large 256px CycleDiffusion 100+100 steps (BASELINE config 3)",
               name="C3: LDM text2img-large-shaped U-Net (context 1280) + KL-f8 VAE, 256x256, custom_steps=99 "
                    "white_box_steps=100 eta=0.1 skip 0, 1 trial, encoder scale 1, decoder CFG 3"),
    "c5": dict(cfg="experiments/bench_afhq_c5.cfg", res=256, batch=4, text=False,
               flop_per_image=(849 + 850 + 100) * 387.9e9,   # 697.8 TFLOP
               metric="images/sec, AFHQ cat->dog 256px, two improved-DDPM U-Nets, reference chain 1000 / 850 / 100 "
                      "(BASELINE config 5)",
               name="C5: two AFHQ improved-DDPM U-Nets (source / target), 256x256, sample_type ddim eta 0.1, "
                    "custom_steps=1000 es_steps=850 refine_steps=100 (translate_afhqcat256_to_afhqdog256_ddim_eta01.cfg)"),
    "c5r": dict(cfg="experiments/bench_afhq_c5_reduced.cfg", res=256, batch=4, text=False,
                flop_per_image=(84 + 85 + 10) * 387.9e9,
                metric="images/sec, AFHQ cat->dog 256px, two improved-DDPM U-Nets, REDUCED chain (BASELINE config 5 / 10)",
                name="C5 REDUCED: two AFHQ improved-DDPM U-Nets (source / target), 256x256, sample_type ddim eta 0.1, "
                     "custom_steps=100 es_steps=85 refine_steps=10 (reference cfg: 1000 / 850 / 100)"),
}


def thread_cpu_seconds():
    """user + system CPU seconds of every thread of this process, by (tid, name) - which thread burns the host cores"""
    out = {}
    tick = os.sysconf("SC_CLK_TCK")
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                with open("/proc/self/task/%s/stat" % tid) as fh:
                    st = fh.read()
                name = st[st.index("(") + 1:st.rindex(")")]
                f = st[st.rindex(")") + 2:].split()
                out[(int(tid), name)] = (int(f[11]) + int(f[12])) / tick
            except (OSError, ValueError):
                pass
    except OSError:
        pass
    return out


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


def cpu_baseline(eng, un, vn, seed=0, quick=False):
    """The CPU oracle (torch fp32 restatement of the reference path, oracle/) timed on this box's host cores on a
    bounded sample of the C2 workload at batch 1: (i) the SD U-Net forward at batch 1 and at batch 2
    (the CFG pair), best of 3 each, one VAE encode and one decode at 512x512, extrapolated linearly to 99 + 99 steps;
    (ii) as the cross-check BASELINE.md asks for, a REAL 5-step DPM-Encoder + 5-step CFG-3 decode loop through
    oracle.samplers, extrapolated the same way. `value` is (i); (ii) is reported beside it. The reference's own
    modules cannot travel to the GPU box; their full 99 + 99 run on 8 cores here took 1333 s / image
    (tests/golden/c2_sd512_e2e.npz: cpu_seconds), i.e. 0.00075 images/s."""
    from oracle import nets, samplers
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
        def best_of(fn, n=1 if quick else 3):
            ts = []
            for _ in range(n):
                t0 = time.time(); fn(); ts.append(time.time() - t0)
            return min(ts), ts

        if not quick:
            nets.openai_unet(usd, ucfg, x[:1], t[:1], ctx[:1])  # warm-up (thread pool, allocator)
        t_u1, l1 = best_of(lambda: nets.openai_unet(usd, ucfg, x[:1], t[:1], ctx[:1]))
        t_u2, l2 = best_of(lambda: nets.openai_unet(usd, ucfg, x, t, ctx))
        img = torch.rand(1, 3, 512, 512, generator=g) * 2 - 1
        t0 = time.time(); mom = nets.vae_encode_moments(vsd, vcfg, img); t_e = time.time() - t0
        t0 = time.time(); nets.vae_decode(vsd, vcfg, mom[:, :4]); t_d = time.time() - t0
        if quick:
            per_img = t_e + N_STEPS * t_u1 + N_STEPS * t_u2 + t_d
            return {"value": 1.0 / per_img, "unit": "images/s", "cores": cores, "kind": "port",
                    "sample": "oracle port: one U-Net forward @B=1 %.2fs + @B=2 %.2fs + VAE enc %.2fs + dec %.2fs, "
                              "extrapolated to 99 + 99 steps (%.0f s/image)" % (t_u1, t_u2, t_e, t_d, per_img)}
        # (ii) a real short loop: 5 encode steps + 5 CFG-3 decode steps of the sampler restatement
        S = 5
        unet = lambda xx, tt, cc: nets.openai_unet(usd, ucfg, xx, tt, cc)
        x0 = mom[:, :4] * 0.18215
        nz = [torch.randn(x0.shape, generator=g) for _ in range(S)]
        t0 = time.time()
        z = samplers.latent_encode(samplers.cfg_model(unet, ctx[:1], ctx[1:], 1.0), x0, S, 0.1, nz, white_box_steps=S + 1)
        t_enc5 = time.time() - t0
        t0 = time.time()
        samplers.latent_decode(samplers.cfg_model(unet, ctx[1:], ctx[:1], 3.0), z[0], torch.stack(z[1:], 1), S, 0.1)
        t_dec5 = time.time() - t0
    per_img = t_e + N_STEPS * t_u1 + N_STEPS * t_u2 + t_d
    per_img_loop = t_e + N_STEPS * (t_enc5 + t_dec5) / S + t_d
    return {"value": 1.0 / per_img, "unit": "images/s", "cores": cores, "kind": "port",
            "value_from_5plus5_loop": 1.0 / per_img_loop,
            "sample": "oracle (torch fp32 CPU), batch 1: U-Net fwd @B=1 best of 3 %.2fs (%s) + @B=2 %.2fs (%s) + VAE enc "
                      "%.2fs + dec %.2fs at 512x512, extrapolated linearly to 99 encode + 99 CFG decode steps (%.0f "
                      "s/image); cross-check: real 5-step encode %.2fs + 5-step CFG decode %.2fs loop -> %.0f s/image"
                      % (t_u1, " ".join("%.2f" % v for v in l1), t_u2, " ".join("%.2f" % v for v in l2), t_e, t_d,
                         per_img, t_enc5, t_dec5, per_img_loop)}


def cpu_baseline_reference():
    """The REFERENCE's own modules (UNetModel / Encoder / Decoder / DDIMSampler.ddpm_ddim_encoding / sample_with_eps,
    BASELINE.md §3) timed on this box's host cores, fp32, batch 1, on a bounded sample of the C2 workload: VAE encode at
    512 x 512, the LAST 3 of the 99 DPM-Encoder steps (skip_steps = 96 on the real 99-step schedule; every step costs
    the same), 3 decode steps with classifier-free guidance 3 (batch 2 forwards), VAE decode; extrapolated linearly to
    99 + 99 steps. The modules come from /root/reference where it is mounted, else from oracle/_ref/ (staged by
    oracle/stage_reference.py at build time, git-ignored, travels with the built library). None if neither exists."""
    from oracle import ref_import
    if not ref_import.available():
        return None
    cores = host_cores()
    torch.set_num_threads(cores)
    with ref_import.session():
        from oracle.gen_golden import RefVAE
        from oracle.gen_golden_full import FULL_VAE, SD_UNET
        from ldm.modules.diffusionmodules.openaimodel import UNetModel
        Sampler = ref_import.ddim_sampler_cls()
        g = torch.Generator().manual_seed(3)
        with torch.no_grad():
            with torch.device("meta"):
                u = UNetModel(**SD_UNET)
                v = RefVAE(FULL_VAE)
            for m in (u, v):  # random weights without the modules' slow default initialisers
                m.to_empty(device="cpu")
                for prm in m.parameters():
                    prm.normal_(0.0, 0.02, generator=g)
                m.eval()
            shim = ref_import.LatentShim(u)
            img = torch.rand(1, 3, 512, 512, generator=g) * 2 - 1
            c_src, c_tgt, uc = (torch.randn(1, 77, 768, generator=g) for _ in range(3))
            t0 = time.time(); x0 = v.moments(img)[:, :4] * 0.18215; t_e = time.time() - t0
            S, skip = 99, 96
            with ref_import.quiet():
                t0 = time.time()
                z = Sampler(shim).ddpm_ddim_encoding(S, batch_size=1, shape=(4, 64, 64), conditioning=c_src, eta=0.1,
                                                     white_box_steps=S + 1, skip_steps=skip, verbose=False, x0=x0,
                                                     unconditional_guidance_scale=1, unconditional_conditioning=uc)
                t_enc = time.time() - t0
                z = torch.stack(z, dim=1)
                n_steps = z.shape[1] - 1
                t0 = time.time()
                x, _ = Sampler(shim).sample_with_eps(S, z[:, 1:], conditioning=c_tgt, batch_size=1, shape=(4, 64, 64),
                                                     eta=0.1, verbose=False, x_T=z[:, 0], skip_steps=skip,
                                                     unconditional_guidance_scale=3.0, unconditional_conditioning=uc)
                t_dec = time.time() - t0
            t0 = time.time(); v.decode(x / 0.18215); t_d = time.time() - t0
    per_img = t_e + S * t_enc / n_steps + S * t_dec / n_steps + t_d
    return {"value": 1.0 / per_img, "unit": "images/s", "cores": cores, "kind": "reference",
            "sample": "the reference's own UNetModel / Encoder / Decoder / DDIMSampler (CPU fp32, batch 1, %s): VAE encode "
                      "%.2fs + %d DPM-Encoder steps %.2fs + %d CFG-3 decode steps %.2fs + VAE decode %.2fs at 512x512, "
                      "extrapolated linearly to 99 + 99 steps (%.0f s/image)"
                      % (ref_import.REF, t_e, n_steps, t_enc, n_steps, t_dec, t_d, per_img)}


def pmc_traffic_per_launch(steps_per_set=16):
    """HBM-side bytes per k_conv_gemm launch from the committed rocprofv3 PMC passes (profiles/README.md):
    FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE over the U-Net forwards of one coalesced C2 launch set (encode at
    B' = 4 x steps, guided decode at twice that), which launch equally often. Returns (bytes, the files it read) - the
    figure is a constant of the committed profile, not a measurement of this run, and the line says which files it came
    from; (None, None) when no profile of this operating point is committed."""
    prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    batches = (4 * steps_per_set, 8 * steps_per_set)
    for rnd in ("r6", "r5", "r4", "r3", "r2b"):  # the newest committed pair
        vals, files = [], []
        for b in batches:
            name = "%s_conv_gemm_traffic_unet_b%d.json" % (rnd, b)
            try:
                with open(os.path.join(prof, name)) as fh:
                    vals.append(float(json.load(fh)["bytes_per_launch"]))
                files.append("profiles/" + name)
            except (OSError, KeyError, ValueError):
                break
        if len(vals) == 2:
            return sum(vals) / len(vals), files
    return None, None


def launch_sets(steps, cap):
    """K queued steps as launch sets of `cap` steps plus one smaller tail set. (Round 5 measured the alternative - sets of equal
    size, 20 steps = 10 + 10 instead of 16 + 4: 3.16 against 3.35-3.44 images/s, GEMM family 833 against 897-929 TFLOP/s. At
    4 images per step a fold of 10 puts 640 row tiles of the 64 x 64 level on 256 CUs - 2.5 rounds - where folds of 4 / 8 / 16
    give 1 / 2 / 4 full rounds: profiles/r5_bench_line_launch_sets_10_10.json.)"""
    if steps <= 0:
        return []
    return [cap] * (steps // cap) + ([steps % cap] if steps % cap else [])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n_ranks, argv):
    """`python bench.py --gpus N` given as ONE command (no WORLD_SIZE in the environment): start the N ranks under
    torch.distributed.run on this node - rendezvous on 127.0.0.1 (the container hostname may not resolve), a free
    port, dmabuf IPC for RCCL - and hand back its exit code; rank 0's JSON line goes to this process's stdout."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n_ranks)))
    rc = 1
    for _attempt in range(3):  # the port is free when probed, not necessarily when torchrun binds it: try another one
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
        t0 = time.time()
        p = subprocess.run(cmd, env=env, cwd=ROOT, stderr=subprocess.PIPE, text=True)
        rc = p.returncode
        sys.stderr.write(p.stderr)
        # retry ONLY a rendezvous that lost the race for its port (fails within seconds, and says so); a bad argument or an
        # import error must not run - and print its error - three times
        lost_port = any(k in p.stderr for k in ("EADDRINUSE", "Address already in use", "address already in use",
                                                "failed to bind", "RendezvousConnectionError", "DistNetworkError"))
        if rc == 0 or time.time() - t0 > 60 or not lost_port:
            break
    return rc


def ranks_seen_by_gather(rank, world, dev):
    """every rank's id through one all-gather on the job's process group: the line proves the collective saw N ranks"""
    if not dist.is_initialized():
        return [rank]
    mine = torch.tensor([rank], device=dev, dtype=torch.int64)
    outs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(outs, mine)
    return [int(o.item()) for o in outs]


def bf16_leg(steps=16):
    """BASELINE.json's C2 line names bf16 storage; the headline runs the fp16 build. The bf16 build of the same
    sources (lib/libcyclediff_bf16.so, built by __graft_entry__.build()) is measured here beside it: one more bench.py process
    on that library (CYCLEDIFF_LIB), `--steps 16` = one full launch set, no other legs. Its parity floor (34 dB against the
    reference's image; the fp16 build holds 50) is what tests/test_gpu_e2e_fullsize.py::
    test_c2_sd_v14_512_end_to_end_on_the_bf16_library runs on the same library."""
    lib = os.path.join(ROOT, "cycle-diffusion_amd", "lib", "libcyclediff_bf16.so")
    if not os.path.exists(lib):
        return {"error": "lib/libcyclediff_bf16.so has not been built (__graft_entry__.build())"}
    env = dict(os.environ, CYCLEDIFF_LIB=lib)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(steps), "--warmup", "0", "--no-cpu-baseline",
           "--no-single-batch", "--no-bf16"]
    try:
        p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
        line = json.loads(p.stdout.strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001 - a side leg never fails the headline
        return {"error": "%s: %s" % (type(e).__name__, e)}
    assert line["dtype"] == "bf16", line["dtype"]
    return {"value": line["value"], "unit": "images/s", "steps": line["steps"], "ms_per_step": line["ms_per_step"],
            "dtype": line["dtype"], "gemm_family_tflops": line["roofline"]["achieved"], "psnr_floor": 34.0,
            "psnr_floor_unit": "dB image PSNR vs the reference's CPU run on the C2 fixture (fp16 build: 50)",
            "psnr_test": "tests/test_gpu_e2e_fullsize.py::test_c2_sd_v14_512_end_to_end_on_the_bf16_library",
            "library": "cycle-diffusion_amd/lib/libcyclediff_bf16.so (-DCD_ACT_FP16=0), same command with CYCLEDIFF_LIB"}


def dry_main(a, wl, rank, world):
    """--dry-run: the host side of a run - rank set-up, contiguous sharding, launch sets, the per-step all-gather in
    step order, barrier-bracketed timing, MAX over ranks, ONE JSON line from rank 0 - with a stand-in for the engine
    call (CPU tensors, gloo instead of RCCL). What the CPU tests drive at world size 2; it measures nothing."""
    from cycle_diffusion_amd.parallel import gather_outputs, run_in_flight, shard_range
    if world > 1 or a.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    B, R = a.batch or wl["batch"], 8
    C = a.coalesce if a.coalesce > 0 else wl.get("coalesce", 4)
    lo, hi = shard_range(B * world, world, rank)
    g = torch.Generator().manual_seed(1)
    images = torch.rand(C, B * world, 3, R, R, generator=g)[:, lo:hi]
    sets = launch_sets(a.steps, C)

    def compute(_r, i):
        n = sets[i] if i < len(sets) else C
        time.sleep(0.001)
        im = images[:n].reshape(n * (hi - lo), 3, R, R)
        return (im, 1.0 - im), torch.zeros(n * (hi - lo)), {}

    seen = []

    def gather(_r, res):
        (orig, img), loss, _ = res
        out = None
        for i in range(img.shape[0] // B):
            sl = slice(i * B, (i + 1) * B)
            out = gather_outputs((orig[sl], img[sl]), loss[sl])
            seen.append(out[0][1].shape[0])
        return out

    def sync():
        if dist.is_initialized():
            dist.barrier()

    done = 0
    while done < a.warmup:
        gather(0, compute(0, len(sets)))
        done += C
    del seen[:]
    sync()
    t0 = time.perf_counter()
    out = run_in_flight(len(sets), 1, compute, gather, pass_index=True)
    sync()
    dt = time.perf_counter() - t0
    if dist.is_initialized():
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ranks_seen = ranks_seen_by_gather(rank, world, torch.device("cpu"))
    res = None
    if rank == 0:
        assert len(seen) == a.steps and all(n == B * world for n in seen), seen  # one full global batch per step
        assert torch.equal(out[0][1], 1.0 - out[0][0])
        res = {"metric": wl["metric"], "value": a.steps * B * world / dt, "unit": "images/s", "n_gpus": world,
               "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "dry run: no engine call, host plumbing only",
               "dry_run": True, "ranks_seen": ranks_seen, "host_threads_per_rank": torch.get_num_threads(),
               "config": {"workload": wl["name"], "batch_per_gpu": B, "global_batch": B * world,
                          "parallelism": "dp%d" % world, "steps_per_launch_set": C,
                          "distributed": "gloo process group" if dist.is_initialized() else "single process"}}
        print(json.dumps(res), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="image triplets per GPU per step (default: the workload's "
                    "BASELINE batch: 4 for C2, README.md:153; 16 for C3, README.md:195)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-batch", action="store_true", help="skip the extra launch-sets-of-one-step measurement")
    ap.add_argument("--no-bf16", action="store_true", help="skip the bf16-library leg (c2 default run on one GPU)")
    ap.add_argument("--single-steps", type=int, default=2, help="steps of the single-batch measurement (c2 default run)")
    ap.add_argument("--coalesce", type=int, default=0,
                    help="steps folded into one engine launch set (same images in flight as that many replicas, ONE "
                         "copy of the weights, that many times the rows per GEMM); 1 = one launch set per step; "
                         "0 = the workload's default (16 for c2 - round 5; rounds 2-4: 8 - and 4 otherwise)")
    ap.add_argument("--in-flight", type=int, default=1,
                    help="launch sets in flight per GPU: independent engine replicas (own HIP stream, workspace and "
                         "weights) driven by host threads")
    ap.add_argument("--precision", default="", help="c2 / c3: fp32x3 or fp32 run the text U-Net in the reference's arithmetic (labelled line; default fp16). c5 / c5r: fp32x3 (default here: the fp32 network with its GroupNorm-fed convolutions as "
                    "three-term split-fp16 GEMMs), fp32 (the reference's own arithmetic, the wrapper's default) or fp16 "
                    "(throughput only: lossy for 'ddim')")
    ap.add_argument("--trials", type=int, default=0, help="c2e only: override n_trials (the unfolded comparison runs a "
                    "1-trial subset: the same 36 candidate chains per trial, one engine call each)")
    ap.add_argument("--no-fold", action="store_true", help="c2e only: one engine call per ensemble member (the "
                    "reference's loop) instead of folding the members that share (skip, scale) into one batch")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group even at world size 1 (exercises the gather path on a 1-GPU box)")
    ap.add_argument("--self-launch", action="store_true",
                    help="start the ranks under torch.distributed.run even at --gpus 1 (what --gpus N > 1 does on its own "
                         "when WORLD_SIZE is unset): the self-launch path on a 1-GPU box; implies --force-dist")
    ap.add_argument("--dry-run", action="store_true",
                    help="host plumbing only (ranks, sharding, gathers, timing, the JSON line) over gloo on CPU tensors, "
                         "no engine call: the CPU tests of the N > 1 launch path")
    a = ap.parse_args()
    wl = WORKLOADS[a.workload]

    if (a.gpus > 1 or a.self_launch) and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a.gpus, sys.argv[1:]))
    a.force_dist = a.force_dist or a.self_launch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # host-side weight synthesis / oracle: stay inside the CPU quota, shared by the ranks of the node
    torch.set_num_threads(max(1, host_cores() // max(1, world)))
    if a.dry_run:
        assert world == a.gpus, "WORLD_SIZE %d does not match --gpus %d" % (world, a.gpus)
        return dry_main(a, wl, rank, world)
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 or world > 1 or a.force_dist:
        assert world == a.gpus, "WORLD_SIZE %d does not match --gpus %d" % (world, a.gpus)
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
    args = get_config(wl["cfg"], config_root=os.path.join(ROOT, "config"))
    if a.workload in ("c5", "c5r") and not a.precision:
        # the split mode holds the same parity floors as the fp32 path (90.7 dB on the full-chain fixture for both) at
        # 2.7x its speed: it is what this bench measures unless `--precision fp32` asks for the reference's own arithmetic
        a.precision = "fp32x3"
    if a.precision:
        # c5 / c5r: fp32x3 (default) | fp32 | fp16 (lossy). c2 / c3 / c2e: fp16 (default, the headline) | fp32x3 | fp32 = the
        # text U-Net and (round 5) the first stage in the reference's arithmetic (`precision = "full"`) - a labelled line beside
        # the headline, never the headline
        args.gan.precision = a.precision
        if a.workload in ("c5", "c5r") and a.precision not in ("fp32", "fp32x3"):
            args.gan.allow_lossy_ddim = True  # throughput-only line: the wrapper refuses 16-bit 'ddim' unless asked by name
    ensemble = a.workload == "c2e"
    if ensemble:
        if a.trials:
            args.gan.n_trials = a.trials
        args.gan.fold_ensemble = not a.no_fold
        args.gan.text_encoder = "clip"  # FrozenCLIPEmbedder on the engine (synthetic weights; hashing tokenizer)
    # Engine replicas: replica r owns stream r, its own engine (workspace, split-K scratch) and weights. With the
    # default coalescing one replica already keeps 64 images in flight (C2); more replicas only overlap kernel tails.
    n_rep = max(1, a.in_flight)
    C = a.coalesce if a.coalesce > 0 else wl.get("coalesce", 4)
    os.environ["CYCLEDIFF_SHARE_SYNTH"] = "1" if n_rep > 1 else "0"  # generate the synthetic weights once per rank
    replicas = []
    for r in range(n_rep):
        st = torch.cuda.Stream(device=dev) if n_rep > 1 else torch.cuda.current_stream(dev)
        with torch.cuda.stream(st):
            torch.manual_seed(0)  # same weights on every rank and replica (main.py:66)
            replicas.append((st, get_model(args.model.name)(args).eval()))
    model = replicas[0][1]
    wrapper = getattr(model, "gan_wrapper", None) or model.source_gan_wrapper
    eng = wrapper.engine
    cda.Engine._SYNTH_CACHE.clear()

    # synthetic batch: global batch = B * world, rank r takes its contiguous slice (ShardSampler, trainer.py:288-293)
    B = a.batch or wl["batch"]
    R = wl["res"]
    lo, hi = shard_range(B * world, world, rank)
    g = torch.Generator().manual_seed(1)
    # C distinct global batches (one per step of a launch set): a folded launch set holds C * B DIFFERENT triplets
    images = torch.rand(C, B * world, 3, R, R, generator=g)[:, lo:hi].to(dev)
    sample_id = torch.arange(lo, hi, device=dev)
    src = [["source prompt %d of step %d" % (i, j) for i in range(lo, hi)] for j in range(C)]
    tgt = [["target prompt %d of step %d" % (i, j) for i in range(lo, hi)] for j in range(C)]
    torch.manual_seed(4 + rank)  # per-rank noise streams
    # launch sets: K steps are issued C at a time (the last set may be smaller)
    sets = launch_sets(a.steps, C)
    S = max(sets)  # steps of the (largest) launch set: the operating point the line is measured at
    single = (not a.no_single_batch) and C > 1 and a.workload == "c2"
    folded = {n: (images[:n].reshape(n * (hi - lo), 3, R, R), sample_id.repeat(n), sum(src[:n], []), sum(tgt[:n], []))
              for n in set(sets) | {1 if single else 0} if n}

    coupled_by_set = {}  # launch-set size (steps) -> did translate() take the coupled loop (None: wrapper has none)

    def compute(r, n):
        """one launch set of n steps (n * B triplets) on replica r"""
        st, m = replicas[r]
        im, sid, s_txt, t_txt = folded[n]
        torch.cuda.set_device(dev)  # the current device is per host thread
        with torch.cuda.stream(st), torch.no_grad():
            if wl["text"]:
                out = m(sample_id=sid, original_image=im, encode_text=s_txt, decode_text=t_txt)
            else:
                out = m(sample_id=sid, original_image=im)
        w_r = getattr(m, "gan_wrapper", None)
        coupled_by_set[n] = getattr(w_r, "last_translate_coupled", None)
        return out

    def gather(r, res):
        """the per-step all-gather (trainer.py:833), once per step of the launch set, in step order"""
        (orig, img), loss, _ = res
        torch.cuda.current_stream(dev).wait_stream(replicas[r][0])
        out = None
        for i in range(img.shape[0] // B):
            sl = slice(i * B, (i + 1) * B)
            out = gather_outputs((orig[sl], img[sl]), loss[sl])
        return out

    def run_sets(sizes):
        """launch sets in order, up to n_rep of them in flight: one host thread per replica computes, then the main
        thread does that round's all-gathers in step order (parallel.run_in_flight)"""
        return run_in_flight(len(sizes), n_rep, lambda r, i: compute(r, sizes[i]), gather, pass_index=True)

    def sync():
        for _st, m in replicas:  # sleep until the engines' streams are idle (cd_engine_synchronize), then the device
            (getattr(m, "gan_wrapper", None) or m.source_gan_wrapper).engine.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize(dev)

    def mini_ensemble():
        """the ensemble's engine calls at their real batch sizes (all decoder scales: the guided ones share launch sets)
        but 4-step chains (skip 95): warm-up (tunes the GEMM shapes of these batch sizes) and the event-instrumented
        roofline sample - a full ensemble is ~10^2 s of GPU per image"""
        keep = wrapper.skip_steps
        wrapper.skip_steps = [95]
        try:
            gather(0, compute(0, 1))
        finally:
            wrapper.skip_steps = keep
        torch.cuda.synchronize(dev)

    # warm-up: every replica runs every launch-set size once (the first one tunes unseen GEMM shapes, the others
    # reuse the table), then whole sets until at least --warmup steps have run
    done = 0
    if ensemble:
        mini_ensemble()
        done = a.warmup
    for r in range(n_rep if not ensemble else 0):
        for n in sorted(folded, reverse=True):
            gather(r, compute(r, n))
            torch.cuda.synchronize(dev)
            done += n
    while done < a.warmup:
        gather(0, compute(0, S))
        done += S
    sync()
    c0 = time.process_time()
    th0 = thread_cpu_seconds()
    t0 = time.perf_counter()
    out = run_sets(sets)  # exactly --steps steps
    sync()
    dt = time.perf_counter() - t0
    host_cpu = time.process_time() - c0  # CPU seconds of this rank (all its threads) over the timed region
    th1 = thread_cpu_seconds()
    busiest = sorted(((th1[k] - th0.get(k, 0.0)) / dt, k[1]) for k in th1)[-3:][::-1]
    if dist.is_initialized():
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(out[0][1]).all()
    ranks_seen = ranks_seen_by_gather(rank, world, dev)

    # BASELINE's literal operating point beside the folded one: launch sets of ONE step (a batch of B triplets per
    # engine call, B' = B through the DPM-Encoder), timed the same way on every rank
    single_dt, single_coupled = None, None
    timed_coupled = {str(n): coupled_by_set.get(n) for n in sorted(set(sets), reverse=True)}  # by launch-set size (steps)
    if single:
        gather(0, compute(0, 1))
        sync()
        t1 = time.perf_counter()
        for _ in range(a.single_steps):
            gather(0, compute(0, 1))
        sync()
        single_dt = time.perf_counter() - t1
        single_coupled = coupled_by_set.get(1)
        if dist.is_initialized():
            tt = torch.tensor([single_dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            single_dt = float(tt.item())

    res = None
    # roofline of the dominant kernel (implicit-GEMM conv / GEMM family): one more launch set (all ranks take part in
    # its gathers) with per-launch HIP events on rank 0's engine stream; achieved = sum(2*M*N*K) / sum(durations)
    if rank == 0:
        eng.prof_enable(True)
    if ensemble:
        mini_ensemble()
    else:
        gather(0, compute(0, S))
    sync()
    if rank == 0:
        ips = a.steps * B * world / dt
        n_launch, k_ms, k_flops = eng.prof_collect()
        eng.prof_enable(False)
        f32 = getattr(wrapper, "precision", "") == "fp32"
        x3 = getattr(wrapper, "precision", "") == "fp32x3"
        # split mode: algorithmic flops (2 M N K of the fp32 conv) against a third of the 16-bit MFMA peak
        peak = PEAK_F32_TFLOPS if f32 else (PEAK_TFLOPS / 3.0 if x3 else PEAK_TFLOPS)
        ach = k_flops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        traffic, traffic_source = pmc_traffic_per_launch(S) if a.workload == "c2" and not (f32 or x3) else (None, None)
        # the ceiling of THIS device on this lease, after the timed region: a bare 16-bit MFMA loop on every CU settles
        # where the package power cap lets it (MI355X, 1400 W: 1.7-1.8 GHz = 1.7-1.8 PFLOP/s)
        sustained = None
        if not f32:
            try:
                s_tf, s_ghz = eng.mfma_sustained(300)
                sustained = {"tflops": s_tf, "ghz": s_ghz, "what": "bare v_mfma_f32_32x32x16 loop on random operands, every "
                             "CU, 0.3 s, after the timed region (csrc/diag.hip)"}
            except Exception as e:  # noqa: BLE001 - a diagnostic never fails the bench line
                sustained = {"error": "%s: %s" % (type(e).__name__, e)}
        fmt = "fp16" if eng.lib.cd_act_format() == 1 else "bf16"
        res = {
            "metric": wl["metric"], "value": ips, "unit": "images/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp32" if f32 else ("fp32 (3 x fp16 split products)" if x3 else fmt), "data": "synthetic",
            "ranks_seen": ranks_seen,  # rank ids through one all-gather on the job's process group
            "config": {"workload": wl["name"], "batch_per_gpu": B, "global_batch": B * world,
                       "parallelism": "dp%d" % world, "steps_per_launch_set": S, "launch_sets": sets,
                       "launch_set_cap": C, "launch_sets_in_flight_per_gpu": n_rep,
                       "images_in_flight_per_gpu": B * S * n_rep,
                       "host_threads_per_rank": torch.get_num_threads(),
                       # encode() + forward() as ONE loop - a forward per step over [encoder rows | decoder rows],
                       # cd_cycle_translate - by launch-set size in steps (the wrapper couples below COUPLE_MAX_TOKENS)
                       "coupled_loop_by_launch_set": timed_coupled,
                       "distributed": "nccl(RCCL) process group" if dist.is_initialized() else "single process",
                       # CPU seconds of this rank (all threads) per wall second of the timed region: the launching thread
                       # sleeps in blocking-sync events (engine step pacing, cd_engine_synchronize)
                       "host_cpu_cores_used": host_cpu / dt,
                       "host_busiest_threads": ["%s %.2f" % (n, c) for c, n in busiest],
                       "storage": ("fp32 activations / weights, v_mfma_f32_32x32x2_f32 (the reference's arithmetic)" if f32
                                   else "fp32 activations / weights; GroupNorm-fed convolutions as hi.wh + lo.wh + hi.wl on "
                                        "v_mfma_f32_32x32x16_f16 (2^-22 per product), the rest on the fp32 path" if x3
                                   else "%s activations / weights, fp32 accumulate (BASELINE.json's C2 line says bf16: same "
                                        "width and MFMA rate; fp16 keeps the DPM-Encoder's 1/sigma amplification 8x smaller, "
                                        "DESIGN.md §5; bf16 is the CD_ACT_FP16=0 build)" % fmt),
                       "weights": wrapper.weights_origin, "flop_per_image": wl["flop_per_image"],
                       # whole_path_frac prices the REFERENCE's work per image; the engine itself launches ~2 % less on
                       # classifier-free-guidance decodes (shared prefix, DESIGN.md 7 r3 f) - roofline.achieved counts
                       # what was launched
                       "flop_per_image_basis": "reference algorithm: every U-Net forward in full"},
            "roofline": {"bound": "mfma", "kernel": "k_conv_f32 (fp32 implicit GEMM)" if f32 else
                         ("k_conv_gemm on split operands (K x 3) + k_conv_f32 for raw-input convs; algorithmic flops, "
                          "peak = 16-bit MFMA peak / 3") if x3 else "k_conv_gemm (all tile instantiations)",
                         "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                         "traffic": traffic, "traffic_unit": "HBM-side bytes per launch (PMC, profiles/)",
                         "traffic_source": traffic_source,
                         "operating_point": ("the ensemble's engine calls at their real batch sizes on 4-step chains (skip 95), "
                                             "per-launch HIP events") if ensemble else
                         "one launch set of %d steps, single stream, per-launch HIP events" % S,
                         "launches_per_step": n_launch / S, "kernel_ms_per_step": k_ms / S,
                         "algorithmic_tflop_per_step": k_flops / 1e12 / S,
                         "whole_path_frac": ips * wl["flop_per_image"] / 1e12 / (world * peak),
                         # what that fraction prices against what: the 16-bit lines against the dense 16-bit MFMA peak; the
                         # fp32 / fp32x3 lines mix three pipes (split products on the 16-bit cores at a third of their rate,
                         # raw-input convs and attention on v_mfma_f32_32x32x2_f32 at 157 TFLOP/s, fp32 VALU norms), so their
                         # fraction is an index against ONE of them, not a utilisation
                         "whole_path_frac_basis": ("algorithmic FLOPs of the reference path / fp32 MFMA peak (157.3 TFLOP/s); "
                                                   "attention and norms run on other pipes" if f32 else
                                                   "algorithmic FLOPs of the reference path / (16-bit MFMA peak / 3): the split "
                                                   "products' ceiling; raw-input convs and attention run on the fp32 MFMA pipe" if x3
                                                   else "algorithmic FLOPs of the reference path / dense 16-bit MFMA peak")},
        }
        if sustained is not None:
            res["roofline"]["sustained_peak"] = sustained
            if sustained.get("tflops"):
                res["roofline"]["frac_of_sustained"] = ach / (sustained["tflops"] / (3.0 if x3 else 1.0))
        if ensemble:
            n_cand = (wrapper.n_trials * len(wrapper.skip_steps) * len(wrapper.encoder_unconditional_guidance_scales) *
                      len(wrapper.decoder_unconditional_guidance_scales))
            full = 15 * 6 * 6
            res["config"]["ensemble"] = {
                "n_trials": wrapper.n_trials, "skip_steps": list(wrapper.skip_steps),
                "decoder_scales": list(wrapper.decoder_unconditional_guidance_scales), "candidates_per_image": n_cand,
                "folded": bool(wrapper.fold_ensemble), "max_fold": wrapper.MAX_FOLD,
                "ranker": type(wrapper.ranker).__name__, "seconds_per_image": dt / (a.steps * B),
                "seconds_per_candidate": dt / (a.steps * B * n_cand)}
            if n_cand != full:  # a subset (--trials): the line's value is per image of THIS subset; scale by candidates
                res["config"]["ensemble"]["images_per_s_at_540_candidates"] = ips * n_cand / full
                res["config"]["flop_per_image"] = wl["flop_per_image"] * n_cand / full
                res["roofline"]["whole_path_frac"] = ips * res["config"]["flop_per_image"] / 1e12 / (world * peak)
        if single_dt is not None:
            sv = a.single_steps * B * world / single_dt
            res["single_batch_value"] = sv  # images/s with ONE batch of B per launch set (`--coalesce 1`)
            res["single_batch"] = {"value": sv, "unit": "images/s", "steps": a.single_steps,
                                   "coupled_loop": single_coupled,
                                   "ms_per_step": 1e3 * single_dt / a.single_steps, "images_in_flight_per_gpu": B,
                                   "whole_path_frac": sv * wl["flop_per_image"] / 1e12 / (world * peak)}
        if (world == 1 and not a.no_bf16 and a.workload == "c2" and not (f32 or x3) and fmt == "fp16"
                and not os.environ.get("CYCLEDIFF_LIB") and not dist.is_initialized()):
            res["bf16"] = bf16_leg()
        if world == 1 and not a.no_cpu_baseline and a.workload == "c2":
            ref = cpu_baseline_reference()
            if ref is not None:  # the reference itself; the oracle port's forward-extrapolated figure beside it
                port = cpu_baseline(eng, wrapper.unet, wrapper.vae, quick=True)
                ref["port_value"], ref["port_sample"] = port["value"], port["sample"]
                res["cpu_baseline"] = ref
            else:
                res["cpu_baseline"] = cpu_baseline(eng, wrapper.unet, wrapper.vae)
        print(json.dumps(res), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return res


if __name__ == "__main__":
    main()
