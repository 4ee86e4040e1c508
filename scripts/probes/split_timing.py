"""Phase timestamps of workgroup 0 of the split-frame MRI kernel (library built with -DLG2S_TIMING).
usage: STEP_AMD_LIB=variants/libstep_amd_timing.so python scripts/probes/split_timing.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from step_amd.forward_operator import get_operator
dev = 'cuda:0'
op = get_operator("mri", acc_factor=6, mode="same_mask", resolution=192, frames=12, num_acs_lines=24)
names = ["tables", "fwd cols", "wait", "load+share", "rows", "inv cols", "sums+arrive", "finish"]
for B, parts in ((1, 6), (1, 3), (1, 2), (2, 6)):
    x = torch.randn(B, 12, 2, 192, 192, device=dev)
    y = op(torch.randn(1, 12, 2, 192, 192, device=dev))
    op.split_frames = parts
    for _ in range(50):
        op.gradient(x, y, return_loss=True)
    torch.cuda.synchronize()
    ws = op._dev_tables[("split_ws", dev)]
    o = 4 * (2 * B * 12 + 2)
    t = ws[o:o + 144].view(torch.int64).cpu().tolist()
    d = [(t[i + 1] - t[i]) / 100.0 for i in range(8)]
    for k in (10,):
        print("   wait: release fence %.2f, add+spin %.2f, acquire fence %.2f" % ((t[k + 1] - t[k]) / 100.0, (t[k + 2] - t[k + 1]) / 100.0, (t[k + 3] - t[k + 2]) / 100.0))
    print("B=%d parts=%d total %.1f us: " % (B, parts, (t[8] - t[0]) / 100.0) + ", ".join("%s %.1f" % (n, v) for n, v in zip(names, d)))
