"""Classifier-free-guidance wrapper — mirror of the reference's `model/cfg_sampler.py:8-31`.

The reference deep-copies y and runs the denoiser twice per step; here the conditional and
unconditional rows are evaluated as ONE 2B-row batch inside the HIP engine and combined
(out_uncond + scale * (out - out_uncond)) in the sampler-update kernel."""
import torch.nn as nn


class ClassifierFreeSampleModel(nn.Module):
    def __init__(self, model):
        super().__init__()
        self.model = model
        assert self.model.cond_mask_prob > 0, \
            "Cannot run a guided diffusion on a model that has not been trained with no conditions"
        self.rot2xyz = self.model.rot2xyz
        self.translation = self.model.translation
        self.njoints = self.model.njoints
        self.nfeats = self.model.nfeats
        self.data_rep = self.model.data_rep
        self.cond_mode = self.model.cond_mode

    def _rgn_bind(self, B, y, device=None, T=None, cache=False):
        assert self.model.cond_mode in ["text", "action"]
        return self.model._rgn_bind(B, y, device, guided=True, T=T, cache=cache)

    def forward(self, x, timesteps, y=None):
        assert self.model.cond_mode in ["text", "action"]
        return self.model(x, timesteps, y, _guided=True)
