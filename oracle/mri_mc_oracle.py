"""TEST INFRASTRUCTURE.  Multi-coil dynamic MRI -- NO REFERENCE COUNTERPART (the reference's
operator is single-coil Cartesian, forward_operator/dynamic_mri.py:42-48; BASELINE.json's
north_star names the coil-sensitivity multiply).  The oracle is the reference's single-coil
formulation applied coil by coil in plain torch on the CPU, with an explicit coil sum:

    y_c = M * view_as_real(fftn(fftshift((x_re + i x_im) * S_c)))       (dynamic_mri.py:44-48 per coil)
    loss = sum_c sum |y_c - yhat_c|^2                                   (base.py:40-41 summed over coils)

and the gradient by torch autograd (base.py:43-49).  Parity of results built on it is UNPINNED by
construction; the pinned case is C = 1, S = 1, which must reproduce oracle/mri_oracle.py."""
import torch

from .mri_oracle import MRIOracle


class MultiCoilMRIOracle(MRIOracle):
    def __init__(self, acc_factor, mode, resolution, frames, num_acs_lines, sens, sigma=0, dtype=torch.float32):
        super().__init__(acc_factor, mode, resolution, frames, num_acs_lines, sigma)
        sens = torch.as_tensor(sens)
        if not torch.is_complex(sens):
            sens = torch.complex(sens[:, 0], sens[:, 1])
        self.cdtype = torch.complex128 if dtype == torch.float64 else torch.complex64
        self.sens = sens.to(self.cdtype)                                   # [C, R, R]

    def __call__(self, x):
        z = torch.complex(x[:, :, 0], x[:, :, 1]).to(self.cdtype)          # [B, T, R, R]
        zc = z.unsqueeze(2) * self.sens                                    # [B, T, C, R, R]
        k = torch.fft.fftn(torch.fft.fftshift(zc, dim=(-2, -1)), dim=(-2, -1))
        y = torch.view_as_real(k).permute(0, 1, 2, 5, 3, 4)                # [B, T, C, 2, R, R]
        return self.mask.to(y.dtype).unsqueeze(2) * y

    def loss(self, x, y):
        return (self(x) - y).pow(2).flatten(1).sum(-1)

    def gradient(self, x, y, return_loss=False):
        xr = x.detach().clone().requires_grad_(True)
        total = self.loss(xr, y).sum()
        (g,) = torch.autograd.grad(total, xr)
        return (g, total.detach()) if return_loss else g
