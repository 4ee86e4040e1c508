import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from step_amd.forward_operator import get_operator
dev = 'cuda:0'
op = get_operator("mri", acc_factor=6, mode="same_mask", resolution=192, frames=12, num_acs_lines=24)
y = op(torch.randn(1, 12, 2, 192, 192, device=dev))
for B in (32, 64, 256):
    x = torch.randn(B, 12, 2, 192, 192, device=dev)
    op.kernel_variant = 2; g2, l2 = op.gradient(x, y, return_loss=True)
    op.kernel_variant = 0; g0, l0 = op.gradient(x, y, return_loss=True)
    print("B", B, "bit-identical to variant 2:", torch.equal(g0, g2), float((l0 - l2).abs() / l2))
    for v in (0, 2):
        op.kernel_variant = v
        for _ in range(30): op.gradient(x, y, return_loss=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): op.gradient(x, y, return_loss=True)
        torch.cuda.synchronize()
        print("  variant", v, "%.4f ms" % ((time.perf_counter() - t0) * 5), flush=True)
