import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from step_amd.forward_operator import get_operator
dev = 'cuda:0'
op = get_operator("mri", acc_factor=6, mode="same_mask", resolution=192, frames=12, num_acs_lines=24)
y = op(torch.randn(1, 12, 2, 192, 192, device=dev))
for B in (64, 256):
    x = torch.randn(B, 12, 2, 192, 192, device=dev)
    for v in [int(a) for a in sys.argv[1:]] or (0, 2, 5):
        op.kernel_variant = v
        for _ in range(30): op.gradient(x, y, return_loss=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): op.gradient(x, y, return_loss=True)
        torch.cuda.synchronize()
        print(os.environ.get("STEP_AMD_LIB", "default")[-28:], "B", B, "variant", v, "%.4f ms" % ((time.perf_counter() - t0) * 10), flush=True)
