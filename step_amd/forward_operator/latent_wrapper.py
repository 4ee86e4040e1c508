"""Latent-space wrapper (reference forward_operator/latent_wrapper.py:5-30): decode with the
stock PyTorch VAE, evaluate the kernel-backed operator loss inside the autograd graph, clamp
the latent gradient to [-1, 1]."""
import inspect

import torch

from .base import Operator


class LatentWrapper(Operator):
    """``op`` seen from latent space through ``model.decode``.  The operator's ``loss`` is a
    torch.autograd.Function whose backward is the fused adjoint kernel, so the only autograd
    work here is the decoder's own backward."""

    def __init__(self, op, model):
        Operator.__init__(self, sigma=op.sigma)
        self.op, self.model = op, model
        # the kernel-backed operators take the hint `unit_upstream` (the loss is only summed, so
        # the gradient arriving at it is one per chain: an operator whose forward already holds
        # d loss / d x hands it on without a scaling pass over the whole video); it travels as an
        # argument into the autograd node, not as state on the shared operator
        try:
            self._hint = "unit_upstream" in inspect.signature(op.loss).parameters
        except (TypeError, ValueError):
            self._hint = False

    def _decode(self, z):
        return self.model.decode(z)

    def __call__(self, z):
        return self.op(self._decode(z))

    def loss(self, z, measurement):
        return self.op.loss(self._decode(z).float(), measurement)          # fp32 into the kernels

    def gradient(self, z, measurement, return_loss=False):
        leaf = z.detach().clone().requires_grad_(True)
        decoded = self._decode(leaf).float()
        if self._hint:
            total = self.op.loss(decoded, measurement, unit_upstream=True).sum()
        else:
            total = self.op.loss(decoded, measurement).sum()
        (g,) = torch.autograd.grad(total, leaf)
        g = g.to(z.dtype).clamp_(min=-1.0, max=1.0)                        # latent_wrapper.py:24-26
        return (g, total) if return_loss else g
