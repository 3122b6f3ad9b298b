"""DirectionalCLIP ranker on the HIP engine: drop-in for model/energy/clean_clip.py:7-41.

`DirectionalCLIPHIP()(img, original_img, encode_text, decode_text) -> (clip_score, dclip_score)`, each [B]: both towers
of OpenAI CLIP ViT-B/32 run on the engine (csrc/clip_text.hip, cd_clip_text_features / cd_clip_image_features);
weights load by the openai/CLIP package's state_dict names (`clip.load("ViT-B/32")`). Host side, as in the reference:
the image preprocessing (Resize 224 bicubic, CenterCrop, Normalize - clip_preprocess minus ToRGB/ToTensor) and the
few-element score arithmetic. `clip.tokenize` needs the package's BPE vocabulary, which is not in this tree: point
CYCLEDIFF_CLIP_TOKENIZER at a directory with vocab.json + merges.txt, otherwise a hash tokenizer (named as the
stand-in it is) keeps the call sites working with clip.tokenize's framing (<|startoftext|> ... <|endoftext|>, zero
padded to 77).
"""
import os

import torch
import torch.nn.functional as F

from ..engine import oclip_text_desc, oclip_vision_desc
from .text_encoders import BOS, EOS, ClipBpeTokenizer, HashTokenizer

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def clip_preprocess(img, res=224):
    """img [B, 3, H, W] in [0, 1] -> [B, 3, res, res]: shorter side to `res` (bicubic, no antialias - torchvision
    0.12 on tensors), centre crop, mean / std normalise."""
    H, W = img.shape[-2:]
    nh, nw = (res, int(res * W / H)) if H <= W else (int(res * H / W), res)
    x = F.interpolate(img, size=(nh, nw), mode="bicubic", align_corners=False)
    t, l = int(round((nh - res) / 2.0)), int(round((nw - res) / 2.0))
    x = x[..., t:t + res, l:l + res]
    mean = torch.tensor(CLIP_MEAN, device=x.device, dtype=x.dtype).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD, device=x.device, dtype=x.dtype).view(1, 3, 1, 1)
    return ((x - mean) / std).contiguous()


class _ClipTokenize:
    """clip.tokenize framing: [sot] tokens [eot], zero padded to the context length."""

    def __init__(self, context_length=77, require_vocab=False):
        vocab_dir = os.environ.get("CYCLEDIFF_CLIP_TOKENIZER", "")
        if require_vocab and not vocab_dir:
            raise FileNotFoundError("CLIP BPE vocabulary missing: point CYCLEDIFF_CLIP_TOKENIZER at a directory with "
                                    "vocab.json + merges.txt (real CLIP weights need the real tokenizer)")
        self.inner = ClipBpeTokenizer(vocab_dir, context_length) if vocab_dir else HashTokenizer(context_length)
        self.context_length = context_length

    def __call__(self, texts):
        ids = self.inner(texts).clone()
        # both inner tokenizers pad with <|endoftext|>; clip.tokenize pads with 0 after the first one
        for b in range(ids.shape[0]):
            first = int((ids[b] == EOS).nonzero()[0])
            ids[b, first + 1:] = 0
        return ids


class DirectionalCLIPHIP:
    def __init__(self, engine, state_dict=None, seed=9, require_vocab=False):
        self.engine = engine
        self.text = engine.create_net(oclip_text_desc())
        self.vision = engine.create_net(oclip_vision_desc())
        if state_dict is not None:
            for net in (self.text, self.vision):
                n, first = engine.load_state_dict(net, state_dict, strict=True)
                if n:
                    raise KeyError("CLIP state_dict lacks %d tensors, first: %s" % (n, first))
            self.weights_origin = "checkpoint"
        else:
            engine.random_init(self.text, seed=seed)
            engine.random_init(self.vision, seed=seed + 1)
            self.weights_origin = "synthetic(seed=%d)" % seed
        self.tokenize = _ClipTokenize(require_vocab=require_vocab)

    def features(self, img=None, text=None):
        if img is not None:
            return self.engine.clip_image_features(self.vision, clip_preprocess(img.to(self.engine.device, torch.float32)))
        return self.engine.clip_text_features(self.text, self.tokenize(text))

    def __call__(self, img, original_img, encode_text, decode_text):
        assert len(decode_text) == img.shape[0] and len(encode_text) == original_img.shape[0]
        n = lambda t: t / t.norm(dim=-1, keepdim=True)
        with torch.no_grad():
            src, tgt = n(self.features(text=encode_text)), n(self.features(text=decode_text))
            im, orig = n(self.features(img=img)), n(self.features(img=original_img))
            clip_score = torch.einsum("bz,bz->b", im, tgt)
            dclip_score = torch.einsum("bz,bz->b", n(im - orig), n(tgt - src))
        return clip_score, dclip_score

    def score_folded(self, img, original_img, encode_text, decode_text, n):
        """The same scores for `n` candidates of one batch concatenated along dim 0 (img: [n * B, ...], candidate-major), with
        the text towers and the source images encoded ONCE and broadcast over the candidates - what a folded ensemble call
        would otherwise recompute n times (the scores are per sample: clean_clip.py:24-31 has no cross-sample term)."""
        bsz = original_img.shape[0]
        assert img.shape[0] == n * bsz and len(encode_text) == bsz and len(decode_text) == bsz
        nrm = lambda t: t / t.norm(dim=-1, keepdim=True)
        with torch.no_grad():
            src, tgt = nrm(self.features(text=list(encode_text))), nrm(self.features(text=list(decode_text)))
            orig = nrm(self.features(img=original_img))
            im = nrm(self.features(img=img))
            src, tgt, orig = src.repeat(n, 1), tgt.repeat(n, 1), orig.repeat(n, 1)
            clip_score = torch.einsum("bz,bz->b", im, tgt)
            dclip_score = torch.einsum("bz,bz->b", nrm(im - orig), nrm(tgt - src))
        return clip_score, dclip_score
