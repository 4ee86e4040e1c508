"""Steady-state phase breakdown of the persistent MRI kernel (workgroup 0, its k-th frame, all CUs busy) from
the probe build scripts/probes/mri_timing_p.hip.  usage: python scripts/probes/mri_phases_p.py lib.so [B] [k]"""
import sys, os, ctypes as C, torch
sys.path.insert(0, '.')
from step_amd.forward_operator import get_operator
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
op = get_operator("mri", acc_factor=6, mode="same_mask", resolution=192, frames=12, num_acs_lines=24)
x = torch.randn(B, 12, 2, 192, 192, device='cuda:0')
y = op(torch.randn(1, 12, 2, 192, 192, device='cuda:0'))
rows, nrows = op._tables(x.device)
part = torch.empty(2 * B * 12, device='cuda:0')
grad = torch.empty_like(x)
L = C.CDLL(os.path.abspath(sys.argv[1]))
st = torch.cuda.current_stream().cuda_stream
names = ['prev loss sum + column FFTs (6 stages)', 'last pass B + unk share + next-frame loads + Rk sync', 'row pass A',
         'row pass B + residual + inv 12', 'inv 16 + sync', 'column IFFTs + stores']
for k in [int(a) for a in sys.argv[3:]] or [3, 6, 9]:
    for _ in range(20):
        rc = L.mri_probe_step(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), B, 12, C.c_void_p(rows.data_ptr()),
                              C.c_void_p(nrows.data_ptr()), op._max_rows, C.c_void_p(part.data_ptr()),
                              C.c_void_p(grad.data_ptr()), k, C.c_void_p(st))
        assert rc == 0
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 16)()
    assert L.mri_probe_timing(out) == 0
    t = list(out)[:7]
    tot = t[6] - t[0]
    print("frame %d of workgroup 0:" % k)
    for i, n in enumerate(names):
        d = t[i + 1] - t[i]
        print('  %-55s %5.1f us  %4.1f %%' % (n, d / 100.0, 100.0 * d / tot))
    print('  total %.1f us' % (tot / 100.0), flush=True)
