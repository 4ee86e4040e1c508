"""Stand-in prior for runs without the pretrained STeP weights (they are remote downloads the
reference expects at configs/task/*.yaml:6-9 and are not available offline).

It exposes the part of ``SpatialTemporalDiffusionModel`` (reference model/__init__.py:78-157) that
the sampling path touches -- ``decode`` / ``encode`` / ``score`` / ``tweedie`` / ``get_latent_shape`` /
``get_data_shape`` / ``parameters`` -- with the documented shapes: latent [T, C, h, w] ->
video [T, C, h f, w f], down-factor f = 8 for the black-hole task and 4 for MRI
(configs/training/*.yaml).  It is a COMPUTE stand-in with random weights: frames are folded into
the batch as in the reference (model/__init__.py:131-134), the decoder is a small strided
transposed-convolution network, the score is that of a Gaussian prior.  Samples drawn with it mean
nothing scientifically; throughput and plumbing are what it is for."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class StandInPrior(nn.Module):
    def __init__(self, video_shape, latent_shape, width=16, seed=0):
        super().__init__()
        self.video_shape, self.latent_shape = tuple(video_shape), tuple(latent_shape)
        T, C, H, W = self.video_shape
        t, c, h, w = self.latent_shape
        assert T == t and H % h == 0 and H // h == W // w, (video_shape, latent_shape)
        self.factor = H // h
        g = torch.Generator().manual_seed(seed)
        ups, ch, f = [], c, self.factor
        while f > 1:                                        # x2 per stage
            ups += [nn.ConvTranspose2d(ch, width, 4, stride=2, padding=1), nn.SiLU()]
            ch, f = width, f // 2
        self.dec = nn.Sequential(*ups, nn.Conv2d(ch, C, 3, padding=1))
        self.enc = nn.Conv2d(C, c, self.factor, stride=self.factor)
        for p in self.parameters():
            with torch.no_grad():
                p.copy_(torch.randn(p.shape, generator=g) * (0.3 / max(p[0].numel(), 1) ** 0.5))

    def get_latent_shape(self):
        return self.latent_shape

    def get_data_shape(self):
        return self.video_shape

    def decode(self, z):
        B, T = z.shape[:2]
        return torch.tanh(self.dec(z.flatten(0, 1))).unflatten(0, (B, T))

    def encode(self, x):
        B, T = x.shape[:2]
        return self.enc(x.flatten(0, 1)).unflatten(0, (B, T))

    def score(self, x, sigma):
        return -x / (1.0 + sigma ** 2)                      # N(0, (1 + sigma^2) I)

    def tweedie(self, x, sigma):
        return x + sigma ** 2 * self.score(x, sigma)
