"""Sampler-level throughput: the whole MCMC inner loop (MCMCSampler.sample: operator gradient + prior
score + noise + fused update per step), not just the operator, at the shipped blackhole settings
(configs/task/blackhole.yaml: hmc, 60 steps, lr 1.2e-5, tau 0.021).  Pixel-space chains with the
'gaussian' prior need no network; --latent adds a stand-in decoder (a strided transposed conv with
the reference's 8x down-factor) through LatentWrapper to show the autograd path.
usage: python scripts/bench_sampler.py [--batch 64] [--steps 60] [--reps 5] [--latent]"""
import argparse, sys, time
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import eht_lite
from step_amd.forward_operator import BlackHoleImagingTorch, LatentWrapper
from step_amd.sampler import MCMCSampler

p = argparse.ArgumentParser()
p.add_argument('--batch', type=int, default=64)
p.add_argument('--steps', type=int, default=60)
p.add_argument('--reps', type=int, default=5)
p.add_argument('--latent', action='store_true')
p.add_argument('--optimize', action='store_true', help='MCMCSampler.optimize (the pixel-space branch of LatentDAPS, daps.py:312)')
p.add_argument('--noise', default='torch', choices=['torch', 'philox'])
p.add_argument('--no-fuse', action='store_true', help='keep the Langevin update out of the adjoint (gradient + mcmc_update_philox)')
p.add_argument('--graph', action='store_true', help='MCMCSampler(graph=True): the loop as replays of one captured step (philox noise)')
a = p.parse_args()
dev = torch.device('cuda:0')
tab = eht_lite.synthetic_table(8, 64, seed=37, sigma_scale=7e-3)
op = BlackHoleImagingTorch(imsize=256, w_cphase=1, w_logcamp=1, w_flux=0.26, loss_normalize=True,
                           ref_multiplier=0.001, device=dev, obs_table=tab)
g = torch.Generator(device=dev).manual_seed(37)
y = op(torch.rand(1, 64, 1, 256, 256, device=dev, generator=g) * 2 - 1)


class StandIn(torch.nn.Module):
    """decode: [B,64,1,32,32] -> [B,64,1,256,256]; score: Gaussian prior around zero."""
    def __init__(self):
        super().__init__()
        self.up = torch.nn.ConvTranspose2d(1, 1, 8, stride=8)
    def decode(self, z):
        b, t = z.shape[:2]
        return torch.tanh(self.up(z.flatten(0, 1))).view(b, t, 1, 256, 256)
    def score(self, z, sigma):
        return -z / (1.0 + sigma ** 2)


mc = MCMCSampler(num_steps=a.steps, lr=1.2e-5, tau=0.021, lr_min_ratio=0.015, prior_solver='gaussian',
                 mc_algo='hmc', momentum=0.0, noise=a.noise, graph=a.graph)
mc.fuse_langevin = not a.no_fuse
if a.noise == 'philox':
    mc.set_chains(seeds=torch.arange(37, 37 + a.batch, dtype=torch.int64, device=dev))
if a.latent:
    model = StandIn().to(dev)
    target, shape = LatentWrapper(op, model), (a.batch, 64, 1, 32, 32)
else:
    model, target, shape = None, op, (a.batch, 64, 1, 256, 256)
x0 = torch.rand(*shape, device=dev, generator=g) * 2 - 1
xt = x0 + 0.1 * torch.randn(*shape, device=dev, generator=g)
# a step size at which the chain stays finite on this random start (the shipped lr belongs to the
# shipped prior's samples; a diverged chain would time the NaN guard, not the loop)
g0 = target.gradient(x0, y)
mc.lr = float(1e-3 * mc.tau ** 2 / g0.abs().max())
del g0
run = (lambda: mc.optimize(x0, target, y, 0.5, 0.0)) if a.optimize else (lambda: mc.sample(xt, model, x0, target, y, 0.5, 0.0))
run()           # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    out = run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.reps
assert torch.isfinite(out).all() and (out != 0).any(), 'the chain diverged (NaN guard returned zeros)'
print(('optimize ' if a.optimize else 'noise=%s%s ' % (a.noise, ' graph' if a.graph else '')) + '%s chains=%d steps=%d: %.2f ms per MCMC step, %.0f chain-steps/s; a 25 x 60-step sample: %.2f s -> %.0f samples/hour/GPU (data-consistency loop only)'
      % ('latent+stand-in decoder' if a.latent else 'pixel space', a.batch, a.steps, 1e3 * dt / a.steps,
         a.batch * a.steps / dt, 25 * dt * 60 / a.steps, 3600 * a.batch / (25 * dt * 60 / a.steps)))
