import sys, os, torch
sys.path.insert(0, '.')
os.environ['STEP_AMD_LIB'] = os.path.abspath(sys.argv[1])
from step_amd.forward_operator import get_operator
dev = 'cuda:0'
op = get_operator("mri", acc_factor=6, mode="same_mask", resolution=192, frames=12, num_acs_lines=24)
for B in (32, 64, 256):
    x = torch.randn(B, 12, 2, 192, 192, device=dev); y = op(torch.randn(1, 12, 2, 192, 192, device=dev))
    for _ in range(3): op.gradient(x, y)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record(); op.gradient(x, y); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev); m = ts[len(ts)//2]
    print('%-14s B=%3d  %.3f ms  %.0f GB/s algorithmic' % (os.path.basename(sys.argv[1]), B, m, 24*12*192*192*B/m/1e6))
