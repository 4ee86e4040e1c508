"""Phase breakdown of the MRI loss+gradient kernel (workgroup 0) from an EXP_MRI_TIMING build.
usage: [STEP_MRI_KERNEL=1] python scripts/mri_phases.py variants/mri_timing.so [B]"""
import sys, os, ctypes as C, torch
sys.path.insert(0, '.')
os.environ['STEP_AMD_LIB'] = os.path.abspath(sys.argv[1])
from step_amd import _lib
from step_amd.forward_operator import get_operator
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
op = get_operator("mri", acc_factor=6, mode="same_mask", resolution=192, frames=12, num_acs_lines=24)
x = torch.randn(B, 12, 2, 192, 192, device='cuda:0')
y = op(torch.randn(1, 12, 2, 192, 192, device='cuda:0'))
for _ in range(20): op.gradient(x, y, return_loss=True)
torch.cuda.synchronize()
L = C.CDLL(os.environ['STEP_AMD_LIB'])
out = (C.c_ulonglong * 16)()
assert L.mri_debug_timing(out) == 0
t = list(out)[:8]
if os.environ.get('STEP_MRI_KERNEL') == '1': t[1] = t[0]
if os.environ.get('STEP_MRI_KERNEL') == '1':
    names = ['load_tables', 'unkept |y|^2', 'column FFTs (6 strips)', 'row FFT fwd', 'residual', 'row FFT inv', 'column IFFTs + stores']
else:
    names = ['column FFTs (6 strips)', 'last pass B + y prefetch', 'row pass A', 'row pass B + residual + inv 12', 'inv 16', 'column IFFTs + stores', 'loss sum']
tot = t[7] - t[0]
for i, n in enumerate(names):
    d = t[i + 1] - t[i]
    print('%-32s %6d ticks (100 MHz) = %5.1f us  %4.1f %%' % (n, d, d / 100.0, 100.0 * d / tot))
print('total %.1f us' % (tot / 100.0))
