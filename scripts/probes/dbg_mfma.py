import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from step_amd.forward_operator import get_operator
dev = 'cuda:0'
for mode, acc_f, acs in (("same_mask", 6, 24), ("split_mask", 8, 12)):
    op = get_operator("mri", acc_factor=acc_f, mode=mode, resolution=192, frames=12, num_acs_lines=acs)
    gen = torch.Generator().manual_seed(5)
    B = 3
    x = torch.randn(B, 12, 2, 192, 192, generator=gen).to(dev)
    y = op(torch.randn(1, 12, 2, 192, 192, generator=gen).to(dev)) + 0.1 * torch.randn(1, 12, 2, 192, 192, generator=gen).to(dev)
    op.split_frames = False
    op.kernel_variant = 2
    g0, l0 = op.gradient(x, y, return_loss=True)
    op.kernel_variant = 5
    g1, l1 = op.gradient(x, y, return_loss=True)
    d = (g1 - g0).abs()
    print(mode, "max|g|", g0.abs().max().item(), "max diff", d.max().item(), "rel", (d.max() / g0.abs().max()).item(), "loss", l0.item(), l1.item(), (l1 - l0).abs().item() / l0.item())
op = get_operator("mri", acc_factor=6, mode="same_mask", resolution=192, frames=12, num_acs_lines=24)
y = op(torch.randn(1, 12, 2, 192, 192, device=dev))
for B in (32, 64, 256):
    x = torch.randn(B, 12, 2, 192, 192, device=dev)
    for v in (0, 5):
        op.kernel_variant = v
        for _ in range(30): op.gradient(x, y, return_loss=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): op.gradient(x, y, return_loss=True)
        torch.cuda.synchronize()
        print("B", B, "variant", v, "%.4f ms" % ((time.perf_counter() - t0) * 10))
