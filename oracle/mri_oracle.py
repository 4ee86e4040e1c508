"""Oracle (TEST INFRASTRUCTURE, not product): PyTorch-CPU restatement of the
dynamic-MRI operator, reference ``forward_operator/dynamic_mri.py`` with the
default ``loss``/``gradient`` of ``forward_operator/base.py``.
Pinned by ``tests/golden/mri_*.npz``.
"""
import numpy as np
import torch


def kept_lines(resolution, acc_factor, num_acs_lines):
    """Sorted k_y rows kept before any split, dynamic_mri.py:19-22."""
    m = np.zeros(resolution)
    m[::acc_factor] = 1
    half = resolution // 2
    m[half - num_acs_lines // 2: half + num_acs_lines // 2] = 1
    return np.where(m)[0]


def build_mask(acc_factor, mode="same_mask", resolution=192, frames=12, num_acs_lines=12):
    """0/1 float32 mask, dynamic_mri.py:9-39.

    same_mask -> [1,1,1,R,1]; split_mask / random_split_mask -> [1,frames,1,R,1]
    (kept rows dealt to frames by ``np.array_split``; the random variant first
    shuffles them with ``np.random.seed(42)`` -- this touches numpy's GLOBAL
    generator exactly as the reference does)."""
    half = resolution // 2
    if mode == "same_mask":
        mask = torch.zeros(1, 1, 1, resolution, 1)
        mask[:, :, :, ::acc_factor, :] = 1
        mask[:, :, :, half - num_acs_lines // 2: half + num_acs_lines // 2, :] = 1
        return mask
    if mode not in ("split_mask", "random_split_mask"):
        raise ValueError("Unknown mode: {}".format(mode))
    if mode == "random_split_mask":
        np.random.seed(42)
    lines = kept_lines(resolution, acc_factor, num_acs_lines)
    if mode == "random_split_mask":
        np.random.shuffle(lines)
    parts = np.array_split(lines, frames)
    mask = torch.zeros(1, frames, 1, resolution, 1)
    for i in range(frames):
        mask[:, i, :, parts[i], :] = 1
    return mask


class MRIOracle:
    def __init__(self, acc_factor, mode="same_mask", resolution=192, frames=12,
                 num_acs_lines=12, sigma=0):
        self.sigma = sigma
        self.mask = build_mask(acc_factor, mode, resolution, frames, num_acs_lines)

    def __call__(self, x):                      # dynamic_mri.py:42-48
        z = x[:, :, 0, ...] + 1j * x[:, :, 1, ...]
        z = torch.fft.fftshift(z, dim=(-2, -1))
        k = torch.fft.fftn(z, dim=(-2, -1))
        k = torch.view_as_real(k).permute(0, 1, 4, 2, 3).contiguous()
        return self.mask.to(x.device) * k

    def measure(self, x):                       # base.py:36-38
        y0 = self(x)
        return y0 + self.sigma * torch.randn_like(y0)

    def loss(self, x, y):                       # base.py:40-41
        return ((self(x) - y) ** 2).flatten(1).sum(-1)

    def gradient(self, x, y, return_loss=False):  # base.py:43-49
        xt = x.clone().detach().requires_grad_(True)
        l = self.loss(xt, y).sum()
        g = torch.autograd.grad(l, xt)[0]
        return (g, l) if return_loss else g
