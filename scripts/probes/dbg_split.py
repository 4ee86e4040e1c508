import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from step_amd.forward_operator import get_operator
dev = 'cuda:0'
op = get_operator("mri", acc_factor=6, mode="same_mask", resolution=192, frames=12, num_acs_lines=24)
gen = torch.Generator().manual_seed(5)
for B, parts in ((1, 6), (1, 3), (1, 2), (4, 3)):
    x = torch.randn(B, 12, 2, 192, 192, generator=gen).to(dev)
    y = op(torch.randn(1, 12, 2, 192, 192, generator=gen).to(dev)) + 0.1 * torch.randn(1, 12, 2, 192, 192, generator=gen).to(dev)
    op.split_frames = False
    g0, l0 = op.gradient(x, y, return_loss=True)
    op.split_frames = parts
    g1, l1 = op.gradient(x, y, return_loss=True)
    g2, l2 = op.gradient(x, y, return_loss=True)
    d = (g1 - g0).abs()
    print("B", B, "parts", parts, "max|g|", g0.abs().max().item(), "max diff", d.max().item(), "n diff", int((d > 0).sum()), "of", d.numel(),
          "rerun equal", torch.equal(g1, g2), "loss", l0.item(), l1.item())
    nz = (d > 0).nonzero()
    if len(nz):
        print("  first diffs", nz[:5].tolist(), "cols with diffs", sorted(set((nz[:, 4] // 32).tolist())), "rows%16", sorted(set((nz[:, 3] % 16).tolist()))[:16],
              "frames", sorted(set(nz[:, 1].tolist())), "planes", sorted(set(nz[:, 2].tolist())))
    ws = op._dev_tables[("split_ws", dev)]
    print("  counters", ws[:4 * (2 * B * 12 + 2)].view(torch.int32).abs().sum().item())
