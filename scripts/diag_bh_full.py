import numpy as np, torch, sys
sys.path.insert(0, '.')
from step_amd import eht_lite
from step_amd.forward_operator import BlackHoleImagingTorch
from tests.helpers import weights_kw
g = dict(np.load('tests/golden/bh_full_n8.npz'))
dev = 'cuda:0'
tab = eht_lite.synthetic_table(8, 64, seed=int(g['seed']), sigma_scale=float(g['sigma_scale']))
gen = torch.Generator().manual_seed(int(g['seed']))
x_gt = torch.rand(1, 64, 1, 256, 256, generator=gen) * 2 - 1
x = torch.rand(1, 64, 1, 256, 256, generator=gen) * 2 - 1
v64 = torch.from_numpy(g['x_obs64_vis']); v32 = torch.from_numpy(g['x_obs_vis'])
print('abs vis: min %.3g med %.3g max %.3g' % (v64.abs().min(), v64.abs().median(), v64.abs().max()))
print('ref32 vs 64: max abs err %.3g, max elementwise rel %.3g' % ((v32-v64).abs().max(), ((v32-v64).abs()/v64.abs()).max()))
for fd in (False, True):
    op = BlackHoleImagingTorch(imsize=256, ref_multiplier=0.001, device=dev, obs_table=tab, force_dense=fd, **weights_kw(g))
    v = op(x.to(dev)).vis.cpu()
    print(op.path, 'vs 64: max abs err %.3g, max elementwise rel %.3g, rms abs err %.3g' % ((v-v64).abs().max(), ((v-v64).abs()/v64.abs()).max(), (v-v64).abs().pow(2).mean().sqrt()))
    # zero-mean input: x (in [-1,1]) directly as x01 w/o offset
    v0 = op.forward_vis(x.to(dev)).cpu()
    A = None
print('ref rms abs err %.3g' % (v32-v64).abs().pow(2).mean().sqrt())
