"""Pair-grid visualizer of the evaluation harness: the reference's `visualizer_program = multi_image`
(visualization/multi_image.py:9-63 over utils/file_utils.py:9-15, i.e. torchvision.utils.save_image with nrow = 8):
(original, translated[, extra]) tuples are interleaved image by image, the first 100 tuples are laid out 8 per row
with a 2-pixel border, once at full size and once bicubically resized to 256 x 256. torchvision is not a
dependency here: the grid layout of torchvision.utils.make_grid (padding 2, pad value 0) is rebuilt directly."""
import math
import os

import torch
import torch.nn.functional as F


def make_grid(images, nrow=8, padding=2, pad_value=0.0):
    """torchvision.utils.make_grid for a [N, C, H, W] batch in [0, 1]."""
    n, c, h, w = images.shape
    xmaps = min(nrow, n)
    ymaps = int(math.ceil(float(n) / xmaps))
    H, W = h + padding, w + padding
    grid = images.new_full((c, H * ymaps + padding, W * xmaps + padding), pad_value)
    k = 0
    for y in range(ymaps):
        for x in range(xmaps):
            if k >= n:
                break
            grid[:, y * H + padding: y * H + padding + h, x * W + padding: x * W + padding + w] = images[k]
            k += 1
    return grid


def save_image(images, path, nrow=8):
    """torchvision.utils.save_image: grid -> *255 + 0.5 -> clamp -> uint8 PNG."""
    from PIL import Image
    grid = make_grid(images.detach().float().cpu(), nrow=nrow)
    arr = grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
    Image.fromarray(arr).save(path)


def visualize(images, description, save_dir, step):
    """Visualizer.visualize: `images` = tuple of 2 or 3 [B, C, H, W] batches (original, translated[, smaller extra])."""
    k = len(images)
    assert k >= 2
    bsz, c, h, w = images[0].shape
    if k == 3 and images[2].shape[-1] != h:
        assert images[2].shape[-1] < h
        images = (images[0], images[1], F.interpolate(images[2], size=(h, w), mode="nearest"))
    stacked = torch.stack([im.detach().float().cpu() for im in images], dim=1).view(bsz * k, c, h, w)[: 100 * k]
    os.makedirs(save_dir, exist_ok=True)
    full = os.path.join(save_dir, "%s_%s.png" % (description, str(step).zfill(6)))
    save_image(stacked, full, nrow=8)
    small = os.path.join(save_dir, "%s_256_%s.png" % (description, str(step).zfill(6)))
    save_image(F.interpolate(stacked, (256, 256), mode="bicubic"), small, nrow=8)
    return full, small
