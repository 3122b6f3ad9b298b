"""Evaluation driver: config + triplet JSON -> translated images + metrics, on the HIP engine.

A compact stand-in for the reference's `main.py --do_eval` (HF Trainer harness, main.py:57-160,
trainer/trainer.py:793-900): same experiment configs (`--cfg experiments/*.cfg`), same model API, same per-rank
sharding (HF ShardSampler: global batches of B * world split contiguously, last one padded by wrap-around) and the same per-image metrics (evaluation/translate_text.py: PSNR, SSIM,
L2 against the input; CLIP / directional CLIP when the config selects `ranker = directional_clip`).

  python main.py --cfg experiments/toy_ddpm_c1.cfg --data triplets.json --output_dir out [--per_device_eval_batch_size 4]
  python -m torch.distributed.run --nproc-per-node 8 main.py ...        # one process per GPU

`--fold N` is the engine's look-ahead: N consecutive dataloader batches of this rank run as ONE model() call (the operating
point bench.py's headline is measured at: 16 batches of 4 = 64 images through the DPM-Encoder, 128 rows through the guided
decode), and the outputs are split back per sample. The reference's driver issues one batch per call
(trainer/trainer.py:788-833 with --per_device_eval_batch_size 4, README.md:153); the samples of a batch are independent, so
folding changes no result beyond kernel summation order. To make that checkable, every SAMPLE owns its noise stream (a
device generator seeded by --seed and the sample id): a run's images do not depend on --fold, on the batch size, on the
number of ranks, or on which rank / batch / slot a sample lands in - including the duplicates the sampler's wrap-around
padding creates (they reproduce the original's image and are not written a second time).
"""
import argparse
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


class SampleStreams:
    """noise_source of the wrappers: one generator per SAMPLE of a (folded) call, seeded by (--seed, sample id); a draw of
    shape [n samples, ...] is the concatenation of each sample's own [1, ...] draw - what that sample draws in any other
    call, batch, slot or rank (round-5 advisor: per-batch streams keyed by the batch's first id made a wrap-around duplicate
    that lands in another batch position draw other noise than its original, and then overwrite it)"""

    def __init__(self, seed, sample_ids, device):
        self.device = device
        self.gens = [torch.Generator(device=device).manual_seed((seed * 1000003 + 7919 * int(i)) % (2 ** 63))
                     for i in sample_ids]

    def __call__(self, shape):
        assert shape[0] == len(self.gens), (shape, len(self.gens))
        return torch.cat([torch.randn((1,) + tuple(shape[1:]), generator=g, device=self.device) for g in self.gens], 0)


def folded_calls(model, batches, fold, seed, device):
    """The look-ahead loop: `batches` (an iterable of collated dataloader batches of this rank, in order; pulled `fold` at a
    time, so a generator keeps host memory at one folded call) run `fold` at a time as ONE model() call; yields (batch dict of
    the folded call, original images, translated images). Every wrapper of the model that has a `noise_source` hook draws
    from the per-sample streams above."""
    import itertools
    wrappers = [w for w in (getattr(model, "gan_wrapper", None), getattr(model, "source_gan_wrapper", None),
                            getattr(model, "target_gan_wrapper", None)) if w is not None]
    fold = max(1, int(fold))
    it = iter(batches)
    while True:
        parts = list(itertools.islice(it, fold))
        if not parts:
            break
        batch = {"sample_id": torch.cat([p["sample_id"] for p in parts]),
                 "original_image": torch.cat([p["original_image"] for p in parts])}
        for k in ("encode_text", "decode_text"):
            if k in parts[0]:
                batch[k] = [t for p in parts for t in p[k]]
        for k in ("is_padding",):  # host-side bookkeeping of the driver: passed through, never handed to the model
            if k in parts[0]:
                batch[k] = [t for p in parts for t in p[k]]
        streams = SampleStreams(seed, batch["sample_id"].tolist(), device)
        for w in wrappers:
            if hasattr(w, "noise_source"):
                w.noise_source = streams
        kw = {"sample_id": batch["sample_id"].to(device), "original_image": batch["original_image"].to(device)}
        if "encode_text" in batch:
            kw.update(encode_text=batch["encode_text"], decode_text=batch["decode_text"])
        with torch.no_grad():
            (orig, img), _loss, _ = model(**kw)
        yield batch, orig, img


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", required=True)
    ap.add_argument("--data", required=True, help="JSON list of {img_path[, encode_text, decode_text]}")
    ap.add_argument("--output_dir", default="output")
    ap.add_argument("--per_device_eval_batch_size", type=int, default=4)
    ap.add_argument("--range", type=int, nargs=2, default=None, metavar=("START", "END"))
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--fold", type=int, default=1,
                    help="look-ahead: run N consecutive batches of this rank as one model() call (larger GEMMs; bench.py's "
                         "headline operating point is --per_device_eval_batch_size 4 --fold 16; under the reference's own Trainer "
                         "the same launch sets are reached with --per_device_eval_batch_size 64); results per image are those of "
                         "--fold 1 up to kernel summation order")
    ap.add_argument("--grid", action="store_true",
                    help="also write the reference's multi_image pair grids (original | translated, 8 per row)")
    ap.add_argument("--synthetic-weights", action="store_true",
                    help="run on seeded synthetic weights when a checkpoint file is missing (default: error, as the "
                         "reference's torch.load); recorded as weights_origin in metrics.json")
    a = ap.parse_args(argv)
    if a.synthetic_weights:
        os.environ["CYCLEDIFF_SYNTHETIC_WEIGHTS"] = "1"

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    import cycle_diffusion_amd  # noqa: F401
    from cycle_diffusion_amd.data.triplets import TripletDataset, collate
    from cycle_diffusion_amd.parallel import shard_indices, shard_padding
    from cycle_diffusion_amd.utils import metrics
    from cycle_diffusion_amd.utils.config_utils import get_config
    from cycle_diffusion_amd.utils.program_utils import get_model

    args = get_config(a.cfg, config_root=os.path.join(ROOT, "config"))
    torch.manual_seed(a.seed)  # same weights / streams on every rank (main.py:66 set_seed)
    model = get_model(args.model.name)(args).eval()
    wrapper = getattr(model, "gan_wrapper", None) or model.source_gan_wrapper
    start, end = a.range if a.range else (0, None)
    ds = TripletDataset(a.data, wrapper.resolution, start, end)
    os.makedirs(a.output_dir, exist_ok=True)
    dev = torch.device("cuda", local)
    rows, grid_pairs = [], []
    import time
    t_start, n_done = time.perf_counter(), 0
    steps = shard_indices(len(ds), a.per_device_eval_batch_size, world, rank)
    pads = shard_padding(len(ds), a.per_device_eval_batch_size, world, rank)

    def batches():  # decoded and collated as the loop pulls them: host memory holds one folded call, not the shard
        for step_idx, pad in zip(steps, pads):
            b = collate([ds[i] for i in step_idx])
            b["is_padding"] = list(pad)
            yield b

    grid_cap = 100 * a.per_device_eval_batch_size  # the reference keeps 100 dataloader batches for its grids
    for batch, orig, img in folded_calls(model, batches(), a.fold, a.seed, dev):
        n_done += img.shape[0]
        if a.grid and rank == 0 and sum(p[0].shape[0] for p in grid_pairs) < grid_cap:
            grid_pairs.append((orig.detach().clamp(0, 1).cpu(), img.detach().clamp(0, 1).cpu()))
        for j in range(img.shape[0]):
            if batch["is_padding"][j]:  # wrap-around padding of the last global batch: its original is written elsewhere
                continue
            o, g = orig[j].clamp(0, 1).cpu(), img[j].clamp(0, 1).cpu()
            sid = int(batch["sample_id"][j])
            row = {"sample_id": sid, "psnr": float(metrics.calculate_psnr(g, o)),
                   "ssim": float(metrics.calculate_ssim(g.permute(1, 2, 0) * 255, o.permute(1, 2, 0) * 255)),
                   "l2": float(metrics.calculate_l2(g, o))}
            for k in ("encode_text", "decode_text"):
                if k in batch:
                    row[k] = batch[k][j]
            rows.append(row)
            from PIL import Image
            Image.fromarray((g.permute(1, 2, 0).numpy() * 255 + 0.5).astype("uint8")).save(
                os.path.join(a.output_dir, "%06d.png" % sid))
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t_start  # translation + per-image metrics + PNG writes of this rank
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, rows)
        rows = [r for part in gathered for r in part]
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and grid_pairs:
        from cycle_diffusion_amd.utils.visualize import visualize
        visualize((torch.cat([p[0] for p in grid_pairs]), torch.cat([p[1] for p in grid_pairs])), "eval", a.output_dir, 0)
    if rank == 0:
        assert len({r["sample_id"] for r in rows}) == len(rows)  # padding rows were skipped where they were produced
        rows.sort(key=lambda r: r["sample_id"])
        summary = {k: sum(r[k] for r in rows) / max(1, len(rows)) for k in ("psnr", "ssim", "l2")}
        with open(os.path.join(a.output_dir, "metrics.json"), "w") as fh:
            json.dump({"summary": summary, "weights_origin": getattr(wrapper, "weights_origin", None),
                       "samples": rows}, fh, indent=1)
        print(json.dumps({"n": len(rows), **summary, "seconds": wall, "images_per_s_this_rank": n_done / wall}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
