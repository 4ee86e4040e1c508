"""Phase breakdown of the pipelined MRI loss+gradient kernel (workgroup 0) from the probe build
scripts/probes/mri_timing.hip (see its header for the hipcc line).
usage: python scripts/probes/mri_phases.py variants/mri_timing.so [B]"""
import sys, os, ctypes as C, torch
sys.path.insert(0, '.')
from step_amd.forward_operator import get_operator
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
op = get_operator("mri", acc_factor=6, mode="same_mask", resolution=192, frames=12, num_acs_lines=24)
x = torch.randn(B, 12, 2, 192, 192, device='cuda:0')
y = op(torch.randn(1, 12, 2, 192, 192, device='cuda:0'))
rows, nrows = op._tables(x.device)
part = torch.empty(2 * B * 12, device='cuda:0')
grad = torch.empty_like(x)
L = C.CDLL(os.path.abspath(sys.argv[1]))
st = torch.cuda.current_stream().cuda_stream
for _ in range(20):
    rc = L.mri_probe_step(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), B, 12, C.c_void_p(rows.data_ptr()),
                          C.c_void_p(nrows.data_ptr()), op._max_rows, C.c_void_p(part.data_ptr()),
                          C.c_void_p(grad.data_ptr()), C.c_void_p(st))
    assert rc == 0
torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
assert L.mri_probe_timing(out) == 0
t = list(out)[:8]
names = ['column FFTs (6 strips)', 'last pass B + y prefetch', 'row pass A', 'row pass B + residual + inv 12',
         'inv 16', 'column IFFTs + stores', 'loss sum']
tot = t[7] - t[0]
for i, n in enumerate(names):
    d = t[i + 1] - t[i]
    print('%-32s %6d ticks (100 MHz) = %5.1f us  %4.1f %%' % (n, d, d / 100.0, 100.0 * d / tot))
print('total %.1f us' % (tot / 100.0))
