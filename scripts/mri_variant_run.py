"""A few MRI gradient steps with a given kernel variant, for rocprofv3 counter passes.
usage: python scripts/mri_variant_run.py VARIANT [B] [mode]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd.forward_operator import get_operator
variant = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
mode = sys.argv[3] if len(sys.argv) > 3 else "same_mask"
op = get_operator("mri", acc_factor=6, mode=mode, resolution=192, frames=12, num_acs_lines=24)
op.kernel_variant = variant
x = torch.randn(B, 12, 2, 192, 192, device='cuda:0'); y = op(torch.randn(1, 12, 2, 192, 192, device='cuda:0'))
for _ in range(3): op.gradient(x, y)
torch.cuda.synchronize()
