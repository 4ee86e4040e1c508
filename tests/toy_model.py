"""Stand-in prior for tests and golden generation (the pretrained STeP priors are
remote downloads).  Exposes the subset of
``SpatialTemporalDiffusionModel`` (reference model/__init__.py:78-157) that the
MCMC path touches: decode / encode / score / get_latent_shape / parameters."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class ToyModel(nn.Module):
    def __init__(self, latent_shape, factor=2, gain=1.5):
        super().__init__()
        self.latent_shape = tuple(latent_shape)
        self.factor, self.gain = factor, gain
        self.dummy = nn.Parameter(torch.zeros(1))

    def get_latent_shape(self):
        return self.latent_shape

    def decode(self, z):
        # [B,T,C,h,w] -> [B,T,C,h*f,w*f]; frames folded into batch like the reference
        B, T = z.shape[:2]
        y = F.interpolate(z.flatten(0, 1), scale_factor=self.factor, mode="nearest")
        return torch.tanh(self.gain * y).unflatten(0, (B, T))

    def encode(self, x):
        B, T = x.shape[:2]
        y = F.avg_pool2d(x.flatten(0, 1), self.factor)
        return (torch.atanh(y.clamp(-0.999, 0.999)) / self.gain).unflatten(0, (B, T))

    def score(self, x, sigma):
        # score of N(0, (1+sigma^2) I)
        return -x / (1 + sigma ** 2)
