"""Any-size MRI kernels (csrc/mri_any.hip) timed next to the tuned 192 x 192 kernels and next to the reference's
formulation on the same GPU (torch.fft = hipFFT + autograd, forward_operator/dynamic_mri.py:42-48 with
base.py:40-49): gradient + loss per call, one measurement for all chains."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd.forward_operator import get_operator  # noqa: E402

dev = "cuda:0"


def timeit(fn, n=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def torch_ref(mask):
    def grad(x, y):
        xt = x.detach().requires_grad_(True)
        z = torch.fft.fftshift(xt[:, :, 0] + 1j * xt[:, :, 1], dim=(-2, -1))
        k = torch.view_as_real(torch.fft.fftn(z, dim=(-2, -1))).permute(0, 1, 4, 2, 3).contiguous()
        l = ((mask * k - y) ** 2).flatten(1).sum(-1).sum()
        return torch.autograd.grad(l, xt)[0], l
    return grad


for res, B in ((192, 256), (192, 64), (192, 4), (128, 256), (256, 64), (256, 256), (320, 64), (384, 64), (512, 32), (200, 64), (97, 64)):
    T = 12
    op = get_operator("mri", acc_factor=6, mode="same_mask", resolution=res, frames=T, num_acs_lines=res // 8)
    x = torch.randn(B, T, 2, res, res, device=dev)
    y = op(torch.randn(1, T, 2, res, res, device=dev))
    line = "R %4d  B %3d  kept rows %3d" % (res, B, op._max_rows)
    if res == 192:
        line += "  tuned %.3f ms" % timeit(lambda: op.gradient(x, y, return_loss=True))
        op.generic_kernels = True
    t_any = timeit(lambda: op.gradient(x, y, return_loss=True))
    nbytes = 16 * T * res * res * B + 8 * T * res * res
    line += "  any-size %.3f ms (%.2f TB/s on x + grad + y)" % (t_any, nbytes / t_any / 1e9)
    ref = torch_ref(op.mask.to(dev))
    g0, _ = ref(x[:2], y)
    g1 = op.gradient(x[:2], y)
    err = ((g0 - g1).abs().max() / g0.abs().max()).item()
    try:
        line += "  reference formulation (torch.fft + autograd) %.3f ms" % timeit(lambda: ref(x, y), n=10, warm=3)
    except RuntimeError as e:                              # out of memory at the largest cases
        line += "  reference formulation: %s" % str(e).split("\n")[0][:60]
    print(line + "  | max rel diff vs it %.1e" % err, flush=True)
