"""Latent-space wrapper (reference forward_operator/latent_wrapper.py:5-30): decode with the
stock PyTorch VAE, evaluate the kernel-backed operator loss inside the autograd graph, clamp
the latent gradient to [-1, 1]."""
import torch

from .base import Operator


class LatentWrapper(Operator):
    def __init__(self, op, model):
        super().__init__(sigma=op.sigma)
        self.op = op
        self.model = model

    def __call__(self, x):
        return self.op(self.model.decode(x))

    def loss(self, pred, observation):
        decoded = self.model.decode(pred)
        return self.op.loss(decoded.float(), observation)

    def gradient(self, pred, observation, return_loss=False):
        pred_tmp = pred.clone().detach().requires_grad_(True)
        loss = self.loss(pred_tmp, observation).sum()
        pred_grad = torch.autograd.grad(loss, pred_tmp)[0]
        pred_grad = torch.clamp(pred_grad.to(pred.dtype), -1, 1)
        if return_loss:
            return pred_grad, loss
        return pred_grad
