"""(image, source text, target text) triplets for evaluation - the reference's DevDataset
(preprocess/translate_text512.py:41-83 over data/translate-text.json: a list of {"img_path", "encode_text",
"decode_text"}): CenterCropLongEdge -> Resize(resolution) -> ToTensor, one item per JSON entry in [start, end).
Unpaired (image-only) entries simply omit the two texts."""
import json
import os

import numpy as np
import torch
from PIL import Image


def center_crop_long_edge(img):
    """utils/transform_utils.py CenterCropLongEdge: square crop of the short-edge size around the centre"""
    w, h = img.size
    s = min(w, h)
    left, top = int(round((w - s) / 2.0)), int(round((h - s) / 2.0))
    return img.crop((left, top, left + s, top + s))


def load_image(path, resolution):
    img = Image.open(path).convert("RGB")  # utils/file_utils.pil_loader
    img = center_crop_long_edge(img)
    if img.size != (resolution, resolution):
        img = img.resize((resolution, resolution), Image.BILINEAR)  # transforms.Resize default for PIL images
    return torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1).contiguous()  # ToTensor


class TripletDataset(torch.utils.data.Dataset):
    def __init__(self, json_path, resolution, start=0, end=None, root=None):
        with open(json_path) as fh:
            raw = json.load(fh)
        self.items = list(enumerate(raw))[start:end]
        self.resolution = resolution
        self.root = root if root is not None else os.path.dirname(os.path.abspath(json_path))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, index):
        idx, meta = self.items[index]
        path = meta["img_path"]
        if not os.path.isabs(path):
            path = os.path.join(self.root, path)
        item = {"sample_id": torch.tensor(idx, dtype=torch.long), "original_image": load_image(path, self.resolution)}
        for k in ("encode_text", "decode_text"):
            if k in meta:
                item[k] = meta[k]
        return item


def collate(items):
    out = {"sample_id": torch.stack([it["sample_id"] for it in items]),
           "original_image": torch.stack([it["original_image"] for it in items])}
    for k in ("encode_text", "decode_text"):
        if k in items[0]:
            out[k] = [it[k] for it in items]
    return out
