import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from step_amd.forward_operator import get_operator
dev = 'cuda:0'
for mode, acc_f, acs, B in (("same_mask", 6, 24, 3), ("split_mask", 8, 12, 50), ("random_split_mask", 6, 24, 70), ("same_mask", 12, 0, 7)):
    op = get_operator("mri", acc_factor=acc_f, mode=mode, resolution=192, frames=12, num_acs_lines=acs)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(B, 12, 2, 192, 192, generator=gen).to(dev)
    y = op(torch.randn(1, 12, 2, 192, 192, generator=gen).to(dev)) + 0.1 * torch.randn(1, 12, 2, 192, 192, generator=gen).to(dev)
    op.split_frames = False
    op.kernel_variant = 2
    g0, l0 = op.gradient(x, y, return_loss=True)
    op.kernel_variant = 6
    g1, l1 = op.gradient(x, y, return_loss=True)
    g2, l2 = op.gradient(x, y, return_loss=True)
    print(mode, B, "equal", torch.equal(g0, g1), torch.equal(g1, g2), "max diff", (g1 - g0).abs().max().item(), "loss rel", ((l1 - l0).abs() / l0).item(), flush=True)
op = get_operator("mri", acc_factor=6, mode="same_mask", resolution=192, frames=12, num_acs_lines=24)
y = op(torch.randn(1, 12, 2, 192, 192, device=dev))
for B in (32, 64, 256):
    x = torch.randn(B, 12, 2, 192, 192, device=dev)
    for v in (0, 6):
        op.kernel_variant = v
        for _ in range(30): op.gradient(x, y, return_loss=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): op.gradient(x, y, return_loss=True)
        torch.cuda.synchronize()
        print("B", B, "variant", v, "%.4f ms" % ((time.perf_counter() - t0) * 10), flush=True)
