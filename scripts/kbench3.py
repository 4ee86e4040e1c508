"""Steady-state timing (200 launches, mean of the last 100) + clocks/power right after the burst."""
import sys, os, subprocess, torch
sys.path.insert(0, '.')
os.environ['STEP_AMD_LIB'] = os.path.abspath(sys.argv[1])
which = sys.argv[2] if len(sys.argv) > 2 else 'fwd'
from step_amd import eht_lite
from step_amd.forward_operator import BlackHoleImagingTorch
dev = 'cuda:0'
tab = eht_lite.synthetic_table(8, 64, seed=37, sigma_scale=7e-3)
op = BlackHoleImagingTorch(imsize=256, ref_multiplier=0.001, device=dev, obs_table=tab, w_flux=0.26,
                           dft=os.environ.get('KB_DFT', 'auto'))
B = 64
x = torch.rand(B, 64, 1, 256, 256, device=dev) * 2 - 1
gv = torch.randn(B, 64 * 28, 2, device=dev); gf = torch.randn(B, 64, device=dev)
fn = (lambda: op._dft(x, 0.5, 0.5, float('-inf'))) if which == 'fwd' else (lambda: op._adjoint(gv, gf, None, 0.5))
n = 2000
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    fn(); ev[i + 1].record()
    if i == n - 200:
        p = subprocess.Popen(['rocm-smi', '--showclocks', '--showpower'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
torch.cuda.synchronize()
ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print(os.path.basename(sys.argv[1]), op.path, which, 'mean[0:50] %.3f  mean[50:200] %.3f  mean[200:1000] %.3f mean[1000:] %.3f' % (
    sum(ts[:50]) / 50, sum(ts[50:200]) / 150, sum(ts[200:1000]) / 800, sum(ts[1000:]) / 1000))
out = p.communicate()[0].decode()
print('\n'.join(l for l in out.splitlines() if 'sclk' in l or 'mclk' in l or 'fclk' in l or 'ower' in l))
