"""Per-component time of one pixel-space MCMC step at B chains (blackhole)."""
import sys, torch, ctypes as C
sys.path.insert(0, '.')
from step_amd import eht_lite
from step_amd.forward_operator import BlackHoleImagingTorch
from step_amd.sampler import MCMCSampler
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device('cuda:0')
tab = eht_lite.synthetic_table(8, 64, seed=37, sigma_scale=7e-3)
op = BlackHoleImagingTorch(imsize=256, w_flux=0.26, ref_multiplier=0.001, device=dev, obs_table=tab)
y = op(torch.rand(1, 64, 1, 256, 256, device=dev) * 2 - 1)
x = torch.rand(B, 64, 1, 256, 256, device=dev) * 2 - 1
xt = x + 0.1 * torch.randn_like(x); prior = (x - xt) / 0.25
mc = MCMCSampler(60, 1.2e-5, tau=0.021, mc_algo='hmc', momentum=0.0)
mc.velocity = torch.randn_like(x)
params = mc._params(1.2e-5, 0.5); flag = torch.zeros(1, device=dev, dtype=torch.int32)
def t(fn, n=30):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
g, _ = op.gradient(x, y, return_loss=True)
eps = torch.randn_like(x)
print('B=%d' % B)
print('operator.gradient   %.3f ms' % t(lambda: op.gradient(x, y, return_loss=True)))
print('torch.randn_like    %.3f ms' % t(lambda: torch.randn_like(x)))
print('fused update        %.3f ms' % t(lambda: mc._fused_update(x, g.view_as(x), xt, prior, eps, params, flag)))
print('x.clone             %.3f ms' % t(lambda: x.clone()))
