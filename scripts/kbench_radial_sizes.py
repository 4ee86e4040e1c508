"""Radial (gridding) operator at resolutions other than 192: gradient + loss per call with the library's own
any-size grid FFT2 (nufft_fft2_any) and with torch.fft (hipFFT) around the same pad / crop kernels."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd.forward_operator import get_operator
dev = "cuda:0"
for R, B in ((128, 64), (256, 32), (96, 64), (104, 64)):
    op = get_operator("mri_radial", spokes=32, resolution=R, frames=12)
    x = torch.randn(B, 12, 2, R, R, device=dev)
    y = op(torch.randn(1, 12, 2, R, R, device=dev))
    line = "R %3d  G %3d  B %2d" % (R, 2 * R, B)
    for route in ("generic", "library"):
        op.grid_fft = route
        for _ in range(10):
            op.gradient(x, y, return_loss=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30):
            op.gradient(x, y, return_loss=True)
        torch.cuda.synchronize()
        line += "  %s %.3f ms" % (route, (time.perf_counter() - t0) / 30 * 1e3)
    print(line, flush=True)
