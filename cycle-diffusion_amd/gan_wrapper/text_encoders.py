"""Text conditioning on the HIP engine: FrozenCLIPEmbedderHIP for Stable Diffusion,
BERTEmbedderHIP for LDM text2img-large.

FrozenCLIPEmbedderHIP mirrors FrozenCLIPEmbedder (model/lib/stable_diffusion/ldm/modules/encoders/modules.py:
136-161): tokenise to 77 ids (`max_length` padding), run the CLIP ViT-L/14 text transformer, return
`last_hidden_state` [B, 77, 768]. The transformer runs on the engine (csrc/clip_text.hip); weights load by the
checkpoint's `cond_stage_model.transformer.text_model.*` names.

The CLIP BPE vocabulary (vocab.json + merges.txt of "openai/clip-vit-large-patch14") is data that is not in this
tree: point CYCLEDIFF_CLIP_TOKENIZER at a directory holding it and transformers' CLIPTokenizer is used, exactly
as the reference does. Without it HashTokenizer - named as the stand-in it is - keeps `list[str]` call sites
working with CLIP's framing (<|startoftext|> words <|endoftext|> padded with <|endoftext|>).
"""
import hashlib
import os

import torch

from ..engine import bert_xtransformer_desc, clip_text_desc

BOS, EOS = 49406, 49407


class HashTokenizer:
    """NOT the CLIP BPE. One stable id in [1, 49405] per lower-cased whitespace-separated word."""

    def __init__(self, max_length=77):
        self.max_length = max_length

    def __call__(self, texts):
        out = torch.full((len(texts), self.max_length), EOS, dtype=torch.int32)
        for b, t in enumerate(texts):
            ids = [BOS]
            for w in t.lower().split()[: self.max_length - 2]:
                ids.append(1 + int.from_bytes(hashlib.sha256(w.encode("utf-8")).digest()[:4], "little") % 49405)
            ids.append(EOS)
            out[b, : len(ids)] = torch.tensor(ids, dtype=torch.int32)
        return out


class ClipBpeTokenizer:
    """transformers' CLIPTokenizer over a local vocabulary directory (modules.py:148-152 call pattern)."""

    def __init__(self, vocab_dir, max_length=77):
        from transformers import CLIPTokenizer
        self.tok = CLIPTokenizer(os.path.join(vocab_dir, "vocab.json"), os.path.join(vocab_dir, "merges.txt"))
        self.max_length = max_length

    def __call__(self, texts):
        enc = self.tok(texts, truncation=True, max_length=self.max_length, return_length=True,
                       return_overflowing_tokens=False, padding="max_length", return_tensors="pt")
        return enc["input_ids"].to(torch.int32)


class FrozenCLIPEmbedderHIP:
    CKPT_PREFIX = "cond_stage_model.transformer."

    def __init__(self, engine, state_dict=None, seed=7, max_length=77, require_vocab=False):
        self.engine = engine
        self.net = engine.create_net(clip_text_desc(positions=max_length))
        if state_dict is not None:
            n, first = engine.load_state_dict(self.net, state_dict, prefix=self.CKPT_PREFIX, strict=True)
            if n:
                raise KeyError("checkpoint lacks %d CLIP text tensors, first: %s" % (n, first))
            self.weights_origin = "checkpoint"
        else:
            engine.random_init(self.net, seed=seed)
            self.weights_origin = "synthetic(seed=%d)" % seed
        vocab_dir = os.environ.get("CYCLEDIFF_CLIP_TOKENIZER", "")
        if require_vocab and not vocab_dir:
            raise FileNotFoundError("CLIP BPE vocabulary missing: point CYCLEDIFF_CLIP_TOKENIZER at a directory with "
                                    "vocab.json + merges.txt (real text-encoder weights need the real tokenizer)")
        self.tokenizer = ClipBpeTokenizer(vocab_dir, max_length) if vocab_dir else HashTokenizer(max_length)

    def __call__(self, texts):
        return self.engine.text_encode(self.net, self.tokenizer(texts))


class BertHashTokenizer:
    """NOT BERT WordPiece. [CLS] words [SEP] padded with [PAD]=0 to max_length (BERTTokenizer, modules.py:47-72:
    truncation, padding="max_length"); one stable id in [1000, 30521] per lower-cased word."""
    CLS, SEP, PAD = 101, 102, 0

    def __init__(self, max_length=77):
        self.max_length = max_length

    def __call__(self, texts):
        out = torch.full((len(texts), self.max_length), self.PAD, dtype=torch.int32)
        for b, t in enumerate(texts):
            ids = [self.CLS]
            for w in t.lower().split()[: self.max_length - 2]:
                ids.append(1000 + int.from_bytes(hashlib.sha256(w.encode("utf-8")).digest()[:4], "little") % 29522)
            ids.append(self.SEP)
            out[b, : len(ids)] = torch.tensor(ids, dtype=torch.int32)
        return out


class BertWordPieceTokenizer:
    """transformers' BertTokenizerFast over a local bert-base-uncased vocabulary (modules.py:53-61 call pattern)."""

    def __init__(self, vocab_dir, max_length=77):
        from transformers import BertTokenizerFast
        self.tok = BertTokenizerFast(os.path.join(vocab_dir, "vocab.txt"))
        self.max_length = max_length

    def __call__(self, texts):
        enc = self.tok(texts, truncation=True, max_length=self.max_length, return_length=True,
                       return_overflowing_tokens=False, padding="max_length", return_tensors="pt")
        return enc["input_ids"].to(torch.int32)


class BERTEmbedderHIP:
    """BERTEmbedder(n_embed=1280, n_layer=32) (model/lib/latentdiff/ldm/modules/encoders/modules.py:75-98): BERT
    tokenizer ids -> x-transformers encoder on the engine -> [B, 77, 1280]."""
    CKPT_PREFIX = "cond_stage_model.transformer."

    def __init__(self, engine, state_dict=None, seed=8, max_length=77, width=1280, layers=32, require_vocab=False):
        self.engine = engine
        self.net = engine.create_net(bert_xtransformer_desc(width=width, layers=layers, positions=max_length))
        if state_dict is not None:
            n, first = engine.load_state_dict(self.net, state_dict, prefix=self.CKPT_PREFIX, strict=True)
            if n:
                raise KeyError("checkpoint lacks %d text-encoder tensors, first: %s" % (n, first))
            self.weights_origin = "checkpoint"
        else:
            engine.random_init(self.net, seed=seed)
            self.weights_origin = "synthetic(seed=%d)" % seed
        vocab_dir = os.environ.get("CYCLEDIFF_BERT_TOKENIZER", "")
        if require_vocab and not vocab_dir:
            raise FileNotFoundError("BERT WordPiece vocabulary missing: point CYCLEDIFF_BERT_TOKENIZER at a directory "
                                    "with vocab.txt (real text-encoder weights need the real tokenizer)")
        self.tokenizer = BertWordPieceTokenizer(vocab_dir, max_length) if vocab_dir else BertHashTokenizer(max_length)

    def __call__(self, texts):
        return self.engine.text_encode(self.net, self.tokenizer(texts))
