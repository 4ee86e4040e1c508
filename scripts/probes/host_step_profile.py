"""Where the host time of one black-hole step goes at one chain (cProfile over 3000 calls)."""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from step_amd import eht_lite
from step_amd.forward_operator import BlackHoleImagingTorch
dev = torch.device('cuda:0')
tab = eht_lite.synthetic_table(8, 64, seed=37, sigma_scale=7e-3)
op = BlackHoleImagingTorch(imsize=256, w_flux=0.26, ref_multiplier=0.001, device=dev, obs_table=tab)
y = op(torch.rand(1, 64, 1, 256, 256, device=dev) * 2 - 1)
x = torch.rand(int(sys.argv[1]) if len(sys.argv) > 1 else 1, 64, 1, 256, 256, device=dev) * 2 - 1
for _ in range(300): op.gradient(x, y, return_loss=True)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3000): op.gradient(x, y, return_loss=True)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
