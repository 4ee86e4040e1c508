"""TEST INFRASTRUCTURE.  Direct non-uniform DFT in fp64 -- the oracle of the gridding kernels
(step_amd/csrc/nufft_grid.hip).  NO REFERENCE COUNTERPART (the reference's MRI operator is
Cartesian, dynamic_mri.py:42-48; pynfft is an install line, README.md:28): parity UNPINNED.

    y[b, t, j] = sum_n z[b, t, n] exp(-2 pi i kappa[t, j] . (n - R/2) / R),   z = x_re + i x_im
    loss[b] = sum |y - yhat|^2 (base.py:40-41 on complex samples),  grad = 2 A^H (A x - yhat)."""
import math

import torch


class NUDFTOracle:
    def __init__(self, kappa, resolution):
        k = torch.as_tensor(kappa, dtype=torch.float64)
        if k.dim() == 2:
            k = k[None]
        n = torch.arange(resolution, dtype=torch.float64) - resolution / 2
        ph = -2 * math.pi / resolution
        self.Ex = torch.polar(torch.ones(1, dtype=torch.float64), ph * k[..., 0, None] * n)    # [Tt, J, R]
        self.Ey = torch.polar(torch.ones(1, dtype=torch.float64), ph * k[..., 1, None] * n)

    def _E(self, T):
        return (self.Ey.expand(T, -1, -1), self.Ex.expand(T, -1, -1))

    def __call__(self, x):
        z = torch.complex(x[:, :, 0].double(), x[:, :, 1].double())                  # [B, T, R, R]
        Ey, Ex = self._E(z.shape[1])
        return torch.view_as_real(torch.einsum("tjy,tjx,btyx->btj", Ey, Ex, z))       # [B, T, J, 2]

    def adjoint(self, v):
        c = torch.view_as_complex(v.double().contiguous())
        Ey, Ex = self._E(c.shape[1])
        z = torch.einsum("tjy,tjx,btj->btyx", Ey.conj(), Ex.conj(), c)
        return torch.stack([z.real, z.imag], dim=2)

    def loss(self, x, y):
        return (self(x) - y.double()).pow(2).flatten(1).sum(-1)

    def gradient(self, x, y, return_loss=False):
        r = self(x) - y.double()
        g = 2 * self.adjoint(r)
        return (g, r.pow(2).sum()) if return_loss else g
