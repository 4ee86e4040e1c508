"""Micro-benchmark of the separable DFT kernels for A/B-ing build variants.
usage: python scripts/kbench.py lib1.so [lib2.so ...]   (each timed in its own process by the caller)"""
import ctypes as C, sys, os, torch
sys.path.insert(0, '.')
lib_path = sys.argv[1]
os.environ['STEP_AMD_LIB'] = lib_path
from step_amd import _lib
_lib.LIB_PATH = os.path.abspath(lib_path)
import numpy as np
from step_amd import eht_lite
from step_amd.forward_operator import BlackHoleImagingTorch
dev = 'cuda:0'
tab = eht_lite.synthetic_table(8, 64, seed=37, sigma_scale=7e-3)
op = BlackHoleImagingTorch(imsize=256, ref_multiplier=0.001, device=dev, obs_table=tab, w_flux=0.26,
                           dft=os.environ.get('KB_DFT', 'auto'))
B = int(os.environ.get('KB_B', '64'))
x = torch.rand(B, 64, 1, 256, 256, device=dev) * 2 - 1
gv = torch.randn(B, 64 * 28, 2, device=dev); gf = torch.randn(B, 64, device=dev)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2], ts[0]
fm, f0 = timeit(lambda: op._dft(x, 0.5, 0.5, float('-inf')))
am, a0 = timeit(lambda: op._adjoint(gv, gf, None, 0.5))
fl = 4.0 * 64 * 28 * 65536 * B
print('%-22s %-5s fwd med %.3f ms (%.1f TF) min %.3f | adj med %.3f ms (%.1f TF) min %.3f' % (os.path.basename(lib_path), op.path, fm, fl / fm / 1e9, f0, am, fl / am / 1e9, a0))
