"""MRI step timing at B = 32, 64, 256 (median of 20 event-timed launches).
usage: python scripts/kbench_mri.py [lib.so] [variant]     (variant 1 pins the first-generation kernel)"""
import sys, os, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    os.environ['STEP_AMD_LIB'] = os.path.abspath(sys.argv[1])
from step_amd.forward_operator import get_operator
dev = 'cuda:0'
tag = os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else 'libstep_amd.so'
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
tag += ' v%d' % variant
for mode, yb in (("same_mask", 1), ("split_mask", 1)) + (() if variant == 4 else (("same_mask", 0),)):
    op = get_operator("mri", acc_factor=6, mode=mode, resolution=192, frames=12, num_acs_lines=24)
    op.kernel_variant = variant
    for B in (32, 64, 256):
        x = torch.randn(B, 12, 2, 192, 192, device=dev); y = op(torch.randn(yb or B, 12, 2, 192, 192, device=dev))
        for _ in range(3): op.gradient(x, y)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in ev:
            a.record(); op.gradient(x, y); b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in ev); m = ts[len(ts)//2]
        print('%-22s %-10s y_batch=%-3d B=%3d  %.3f ms  %.0f GB/s algorithmic' % (tag, mode, yb or B, B, m, 24*12*192*192*B/m/1e6), flush=True)
