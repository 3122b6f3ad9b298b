"""Model API (text-guided): same contract as model/text_unsupervised_translation.py:9-47 —
Model(args); forward(sample_id, original_image, encode_text, decode_text) ->
((original_image, img), zeros_like(sample_id).float(), {})."""
import torch
import torch.nn as nn

from ..gan_wrapper.get_gan_wrapper import get_gan_wrapper


class TextUnsupervisedTranslation(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.gan_wrapper = get_gan_wrapper(args.gan)

    def forward(self, sample_id, original_image, encode_text, decode_text):
        self.gan_wrapper.eval()
        assert not self.training
        if hasattr(self.gan_wrapper, "translate"):
            # encode() + forward() as the engine's coupled loop: ONE U-Net forward per step over [encoder rows | decoder
            # rows] (include/cyclediff.h cd_cycle_translate); same draws, member order and per-sample arithmetic
            img = self.gan_wrapper.translate(original_image, encode_text, decode_text)
        else:
            z_ensemble = self.gan_wrapper.encode(image=original_image, encode_text=encode_text)
            img = self.gan_wrapper(z_ensemble=z_ensemble, original_img=original_image, encode_text=encode_text,
                                   decode_text=decode_text)
        return (original_image, img), torch.zeros_like(sample_id).float(), dict()

    @property
    def device(self):
        return next(self.parameters()).device


Model = TextUnsupervisedTranslation
