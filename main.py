"""Evaluation driver: config + triplet JSON -> translated images + metrics, on the HIP engine.

A compact stand-in for the reference's `main.py --do_eval` (HF Trainer harness, main.py:57-160,
trainer/trainer.py:793-900): same experiment configs (`--cfg experiments/*.cfg`), same model API, same per-rank
sharding (HF ShardSampler: global batches of B * world split contiguously, last one padded by wrap-around) and the same per-image metrics (evaluation/translate_text.py: PSNR, SSIM,
L2 against the input; CLIP / directional CLIP when the config selects `ranker = directional_clip`).

  python main.py --cfg experiments/toy_ddpm_c1.cfg --data triplets.json --output_dir out [--per_device_eval_batch_size 4]
  python -m torch.distributed.run --nproc-per-node 8 main.py ...        # one process per GPU
"""
import argparse
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", required=True)
    ap.add_argument("--data", required=True, help="JSON list of {img_path[, encode_text, decode_text]}")
    ap.add_argument("--output_dir", default="output")
    ap.add_argument("--per_device_eval_batch_size", type=int, default=4)
    ap.add_argument("--range", type=int, nargs=2, default=None, metavar=("START", "END"))
    ap.add_argument("--seed", type=int, default=42)
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
    from cycle_diffusion_amd.parallel import shard_indices
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
    for step_idx in shard_indices(len(ds), a.per_device_eval_batch_size, world, rank):
        batch = collate([ds[i] for i in step_idx])
        kw = {"sample_id": batch["sample_id"].to(dev), "original_image": batch["original_image"].to(dev)}
        if "encode_text" in batch:
            kw.update(encode_text=batch["encode_text"], decode_text=batch["decode_text"])
        with torch.no_grad():
            (orig, img), _loss, _ = model(**kw)
        n_done += img.shape[0]
        if a.grid and rank == 0 and len(grid_pairs) < 100:
            grid_pairs.append((orig.detach().clamp(0, 1).cpu(), img.detach().clamp(0, 1).cpu()))
        for j in range(img.shape[0]):
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
        rows = list({r["sample_id"]: r for r in rows}.values())  # the sampler's wrap-around padding repeats samples
        rows.sort(key=lambda r: r["sample_id"])
        summary = {k: sum(r[k] for r in rows) / max(1, len(rows)) for k in ("psnr", "ssim", "l2")}
        with open(os.path.join(a.output_dir, "metrics.json"), "w") as fh:
            json.dump({"summary": summary, "weights_origin": getattr(wrapper, "weights_origin", None),
                       "samples": rows}, fh, indent=1)
        print(json.dumps({"n": len(rows), **summary, "seconds": wall, "images_per_s_this_rank": n_done / wall}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
