"""Operator registry and base class with the reference's interface
(reference forward_operator/base.py:8-55): same registry names, same method
signatures, same error behaviour (re-registration warns and overwrites, an
unknown name raises NameError)."""
from abc import ABC, abstractmethod

import torch

from .._registry import Registry

__all__ = ["Operator", "register_operator", "get_operator"]

__OPERATOR__ = Registry("operator")
register_operator = __OPERATOR__.register
get_operator = __OPERATOR__.create


class Operator(ABC):
    """y = A(x) + noise.  Interface of the reference's base class (forward_operator/base.py:28-55):
    ``__call__`` is the noiseless operator; the generic ``loss`` / ``gradient`` below go through
    torch autograd and are what the kernel-backed subclasses replace with fused launches while
    keeping the same return values."""

    def __init__(self, sigma=0.05):
        self.sigma = sigma                      # measurement noise level, read by LatentWrapper

    @abstractmethod
    def __call__(self, x):
        """Noiseless measurement A(x)."""

    def measure(self, x):
        """A(x) plus white Gaussian noise of standard deviation ``sigma``."""
        clean = self(x)
        return clean + torch.randn_like(clean) * self.sigma

    def loss(self, x, y):
        """Squared residual per chain: [B, ...] -> [B]."""
        residual = self(x) - y
        return residual.pow(2).flatten(start_dim=1).sum(dim=-1)

    def gradient(self, x, y, return_loss=False):
        """d/dx of the loss summed over the batch; with ``return_loss`` also that scalar."""
        leaf = x.detach().clone().requires_grad_(True)
        total = self.loss(leaf, y).sum()
        (grad,) = torch.autograd.grad(total, leaf)
        return (grad, total) if return_loss else grad

    def log_likelihood(self, x, y):
        return self.loss(x, y) / (-2.0 * self.sigma ** 2)

    def likelihood(self, x, y):
        return self.log_likelihood(x, y).exp()
