"""Timing distribution of the separable DFT kernels: per-launch event times in launch order and
the mean over a back-to-back batch.  usage: python scripts/kbench2.py lib.so"""
import sys, os, torch
sys.path.insert(0, '.')
os.environ['STEP_AMD_LIB'] = os.path.abspath(sys.argv[1])
from step_amd import eht_lite
from step_amd.forward_operator import BlackHoleImagingTorch
dev = 'cuda:0'
tab = eht_lite.synthetic_table(8, 64, seed=37, sigma_scale=7e-3)
op = BlackHoleImagingTorch(imsize=256, ref_multiplier=0.001, device=dev, obs_table=tab, w_flux=0.26,
                           dft=os.environ.get('KB_DFT', 'auto'))
B = int(os.environ.get('KB_B', '64'))
x = torch.rand(B, 64, 1, 256, 256, device=dev) * 2 - 1
gv = torch.randn(B, 64 * 28, 2, device=dev); gf = torch.randn(B, 64, device=dev)
def dist(fn, n=40):
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    per = [a.elapsed_time(b) for a, b in ev]
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return per, a.elapsed_time(b) / n
for name, fn in (('fwd', lambda: op._dft(x, 0.5, 0.5, float('-inf'))), ('adj', lambda: op._adjoint(gv, gf, None, 0.5))):
    per, mean = dist(fn)
    print(os.path.basename(sys.argv[1]), op.path, name, 'batch mean %.3f ms | per-launch:' % mean, ' '.join('%.2f' % p for p in per))
