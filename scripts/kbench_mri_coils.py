"""Multi-coil MRI step: the three routes (one kernel / split into three kernels / first-generation coil loop)
over coil counts and batches.  usage: python scripts/kbench_mri_coils.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd.forward_operator import get_operator
dev = 'cuda:0'
for C in (1, 2, 4, 8, 16):
    op = get_operator("mri_multicoil", coils=C, acc_factor=6, mode="same_mask", resolution=192, frames=12, num_acs_lines=24)
    y = op(torch.randn(1, 12, 2, 192, 192, device=dev))
    for B in (4, 16, 64):
        x = torch.randn(B, 12, 2, 192, 192, device=dev)
        res = []
        for name, fused, split in (("fused", True, None), ("split", False, True), ("loop", False, False)):
            op.fused_coils, op.split_coils = fused, split
            for _ in range(10): op.gradient(x, y, return_loss=True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 30
            for _ in range(n): op.gradient(x, y, return_loss=True)
            torch.cuda.synchronize()
            res.append("%s %.3f ms" % (name, (time.perf_counter() - t0) / n * 1e3))
        print("C=%d B=%d: " % (C, B) + ", ".join(res), flush=True)
