"""Latent-space wrapper (reference forward_operator/latent_wrapper.py:5-30): decode with the
stock PyTorch VAE, evaluate the kernel-backed operator loss inside the autograd graph, clamp
the latent gradient to [-1, 1]."""
import contextlib

import torch

from .base import Operator


@contextlib.contextmanager
def _unit_upstream(op):
    """While active, ``op.loss``'s backward may assume that the gradient arriving from upstream
    is exactly one per chain (the loss is only summed): an operator whose forward already holds
    d loss / d x then hands it on without a scaling pass over the whole video."""
    op._unit_upstream = True
    try:
        yield
    finally:
        op._unit_upstream = False


class LatentWrapper(Operator):
    """``op`` seen from latent space through ``model.decode``.  The operator's ``loss`` is a
    torch.autograd.Function whose backward is the fused adjoint kernel, so the only autograd
    work here is the decoder's own backward."""

    def __init__(self, op, model):
        Operator.__init__(self, sigma=op.sigma)
        self.op, self.model = op, model

    def _decode(self, z):
        return self.model.decode(z)

    def __call__(self, z):
        return self.op(self._decode(z))

    def loss(self, z, measurement):
        return self.op.loss(self._decode(z).float(), measurement)          # fp32 into the kernels

    def gradient(self, z, measurement, return_loss=False):
        leaf = z.detach().clone().requires_grad_(True)
        total = self.loss(leaf, measurement).sum()
        with _unit_upstream(self.op):
            (g,) = torch.autograd.grad(total, leaf)
        g = g.to(z.dtype).clamp_(min=-1.0, max=1.0)                        # latent_wrapper.py:24-26
        return (g, total) if return_loss else g
