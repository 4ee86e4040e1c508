"""MRI step at small batches: eager wall clock per step (return_loss=True) and host profile.
usage: python scripts/kbench_mri_small.py [B ...]"""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd.forward_operator import get_operator
dev = 'cuda:0'
op = get_operator("mri", acc_factor=6, mode="same_mask", resolution=192, frames=12, num_acs_lines=24)
y = op(torch.randn(1, 12, 2, 192, 192, device=dev))
for B in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 21, 32]:
    x = torch.randn(B, 12, 2, 192, 192, device=dev)
    for _ in range(200): op.gradient(x, y, return_loss=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 1000
    for _ in range(n): op.gradient(x, y, return_loss=True)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / n * 1e6
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record(); op.gradient(x, y, return_loss=True); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    print('B=%-3d eager %.1f us per step (%.0f k chain-steps/s), event-timed %.1f us' % (B, us, B / us * 1e3, ts[10] * 1e3), flush=True)
x = torch.randn(1, 12, 2, 192, 192, device=dev)
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): op.gradient(x, y, return_loss=True)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(10)
