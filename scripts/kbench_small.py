"""Black-hole step at small batches: per-kernel event times, eager step, graph-replayed step.
usage: python scripts/kbench_small.py [B ...]"""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import eht_lite
from step_amd.forward_operator import BlackHoleImagingTorch
dev = torch.device('cuda:0')
tab = eht_lite.synthetic_table(8, 64, seed=37, sigma_scale=7e-3)
op = BlackHoleImagingTorch(imsize=256, w_flux=0.26, ref_multiplier=0.001, device=dev, obs_table=tab)
import os
op.fused_step = os.environ.get('FUSED', '1') == '1'
y = op(torch.rand(1, 64, 1, 256, 256, device=dev) * 2 - 1)
for B in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16, 64]:
    x = torch.rand(B, 64, 1, 256, 256, device=dev) * 2 - 1
    for _ in range(200): op.gradient(x, y, return_loss=True)
    torch.cuda.synchronize()
    prof = {}
    op._prof = prof
    for _ in range(50): op.gradient(x, y, return_loss=True)
    torch.cuda.synchronize()
    op._prof = None
    med = {k: sorted(a.elapsed_time(b) for a, b in v)[len(v) // 2] * 1e3 for k, v in prof.items()}
    n = 500
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): op.gradient(x, y, return_loss=True)
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / n * 1e6
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        op.gradient(x, y, return_loss=True)
        with torch.cuda.graph(g, stream=s):
            out = op.gradient(x, y, return_loss=True)
    torch.cuda.synchronize()
    for _ in range(50): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / n * 1e6
    print('B=%-3d kernels us %s | eager step %.1f us | graph step %.1f us | %.0f k chain-steps/s (graph)'
          % (B, ' '.join('%s=%.1f' % kv for kv in med.items()), eager, graph, B / graph * 1e3), flush=True)
