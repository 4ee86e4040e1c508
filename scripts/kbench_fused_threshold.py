"""Black-hole step, two-launch (fused closure stage) against three-kernel path per batch size.
usage: python scripts/kbench_fused_threshold.py [B ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import eht_lite
from step_amd.forward_operator import BlackHoleImagingTorch
dev = torch.device('cuda:0')
tab = eht_lite.synthetic_table(8, 64, seed=37, sigma_scale=7e-3)
op = BlackHoleImagingTorch(imsize=256, w_flux=0.26, ref_multiplier=0.001, device=dev, obs_table=tab)
y = op(torch.rand(1, 64, 1, 256, 256, device=dev) * 2 - 1)
for B in [int(a) for a in sys.argv[1:]] or [16, 24, 32, 48, 64]:
    x = torch.rand(B, 64, 1, 256, 256, device=dev) * 2 - 1
    res = []
    for rep in range(2):
        for lim in (10 ** 6, 0):
            op.fused_max_batch = lim
            for _ in range(100): op.gradient(x, y, return_loss=True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 400
            for _ in range(n): op.gradient(x, y, return_loss=True)
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / n * 1e6)
    print('B=%-3d two-launch %.1f / %.1f us, three-kernel %.1f / %.1f us' % (B, res[0], res[2], res[1], res[3]), flush=True)
