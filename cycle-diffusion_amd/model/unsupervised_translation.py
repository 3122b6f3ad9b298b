"""Model API (unpaired, two pixel DDPMs): same contract as model/unsupervised_translation.py:9-62."""
import torch
import torch.nn as nn

from ..gan_wrapper.get_gan_wrapper import get_gan_wrapper


class UnsupervisedTranslation(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.source_gan_wrapper = get_gan_wrapper(args.gan)
        self.target_gan_wrapper = get_gan_wrapper(args.gan, target=True)
        assert self.source_gan_wrapper.resolution == self.target_gan_wrapper.resolution

    def forward(self, sample_id, class_label=None, original_image=None):
        self.source_gan_wrapper.eval()
        self.target_gan_wrapper.eval()
        assert not self.training
        if getattr(self.source_gan_wrapper, "model_embedding_space", False):
            raise NotImplementedError()
        if getattr(self.source_gan_wrapper, "enforce_class_input", False):
            assert getattr(self.target_gan_wrapper, "enforce_class_input", False)
            assert class_label is not None
            z = self.source_gan_wrapper.encode(image=original_image, class_label=class_label)
            img = self.target_gan_wrapper(z=z, class_label=class_label)
        else:
            assert class_label is None
            z = self.source_gan_wrapper.encode(image=original_image)
            img = self.target_gan_wrapper(z=z)
        return (original_image, img), torch.zeros_like(sample_id).float(), dict()

    @property
    def device(self):
        return next(self.parameters()).device


Model = UnsupervisedTranslation
