"""Where a small-batch sampler step spends its time: operator step, update kernel, both (wall clock per call)."""
import os, sys, time, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from step_amd import eht_lite, _lib
from step_amd.forward_operator import BlackHoleImagingTorch
from step_amd.sampler import MCMCSampler
dev = torch.device('cuda:0')
tab = eht_lite.synthetic_table(8, 64, seed=37, sigma_scale=7e-3)
op = BlackHoleImagingTorch(imsize=256, w_cphase=1, w_logcamp=1, w_flux=0.26, loss_normalize=True, ref_multiplier=0.001, device=dev, obs_table=tab)
y = op(torch.rand(1, 64, 1, 256, 256, device=dev) * 2 - 1)
for B in (1, 8):
    x = torch.rand(B, 64, 1, 256, 256, device=dev) * 2 - 1
    xt = x.clone(); prior = torch.zeros_like(x); grad = torch.zeros_like(x)
    mc = MCMCSampler(num_steps=60, lr=1.2e-5, tau=0.021, prior_solver='gaussian', mc_algo='hmc', momentum=0.0, noise='philox')
    params = mc._params(1e-5, 0.5)
    seeds = torch.arange(B, dtype=torch.int64, device=dev); flag = torch.zeros(1, dtype=torch.int32, device=dev)
    def upd():
        rc = _lib.lib().mcmc_update_philox(x.data_ptr(), None, grad.data_ptr(), xt.data_ptr(), prior.data_ptr(), x.numel(),
                                           C.byref(params), seeds.data_ptr(), x[0].numel(), 3, None, 0, None, flag.data_ptr(), _lib.stream_of(x))
    def both():
        g = op.gradient(x, y, return_loss=True); upd()
    for name, fn in (("op.gradient", lambda: op.gradient(x, y, return_loss=True)), ("update kernel", upd), ("both", both)):
        for _ in range(50): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(300): fn()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print('B=%d %-14s host issue %.1f us/call, with GPU drain %.1f us/call' % (B, name, (t1 - t0) / 300 * 1e6, (t2 - t0) / 300 * 1e6), flush=True)
    mc.set_chains(seeds=seeds)
    mc.sample(xt, None, x, op, y, 0.5, 0.0); torch.cuda.synchronize(); t0 = time.perf_counter()
    mc.sample(xt, None, x, op, y, 0.5, 0.0); torch.cuda.synchronize()
    print('B=%d sample(): %.1f us/step' % (B, (time.perf_counter() - t0) / 60 * 1e6), flush=True)
