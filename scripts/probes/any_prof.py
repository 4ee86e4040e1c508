import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from step_amd.forward_operator import get_operator
res, B = int(sys.argv[1]), int(sys.argv[2])
op = get_operator("mri", acc_factor=6, mode="same_mask", resolution=res, frames=12, num_acs_lines=res // 8)
op.generic_kernels = True
x = torch.randn(B, 12, 2, res, res, device="cuda:0")
y = op(torch.randn(1, 12, 2, res, res, device="cuda:0"))
for _ in range(20):
    op.gradient(x, y, return_loss=True)
torch.cuda.synchronize()
