"""GPU tests of the chain semantics of the sharded driver (step_amd/driver.py) and of the
in-kernel Philox noise (mcmc_update_philox) behind it."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle.philox_oracle import chain_noise
from tests.helpers import assert_close
from tests.toy_model import ToyModel

pytestmark = pytest.mark.gpu


def test_philox_kernel_matches_the_oracle(dev):
    from step_amd import _lib
    B, L = 3, 4 * 8191
    seeds = torch.tensor([37, 38, 2 ** 40 + 5], dtype=torch.int64, device=dev)
    gen = torch.Generator(device=dev).manual_seed(1)
    x, g, xt, pr, vel = (torch.randn(B, L, device=dev, generator=gen) for _ in range(5))
    x0, v0 = x.clone(), vel.clone()
    eps = torch.empty_like(x)
    p = _lib.mcmc_params_t(0.7, 0.3, 1e-2, 0.1, 0.1414, 0.6, 0.9, 1)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    step = 2 ** 32 + 7                                         # the high word of the counter as well
    rc = _lib.lib().mcmc_update_philox(x.data_ptr(), vel.data_ptr(), g.data_ptr(), xt.data_ptr(), pr.data_ptr(),
                                       x.numel(), C.byref(p), seeds.data_ptr(), L, step, None, 0, eps.data_ptr(),
                                       flag.data_ptr(), _lib.stream_of(x))
    _lib.check(rc, "mcmc_update_philox")
    want = np.stack([chain_noise(int(s), step, L) for s in seeds.tolist()])
    got = eps.cpu().numpy()
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()
    f = lambda v: torch.tensor(v, dtype=torch.float32).item()
    s = -g * f(0.7) + (xt - x0) * f(0.3) + pr
    v = f(0.6) * v0 + f(0.1) * s + f(0.9) * eps
    assert (vel - v).abs().max().item() <= 1e-5 and (x - (x0 + v * f(0.1))).abs().max().item() <= 1e-5
    # the device-side step counter gives the same noise (graph replay)
    x2, vel2, eps2 = x0.clone(), v0.clone(), torch.empty_like(x)
    ctr = torch.tensor([step], dtype=torch.int64, device=dev)
    rc = _lib.lib().mcmc_update_philox(x2.data_ptr(), vel2.data_ptr(), g.data_ptr(), xt.data_ptr(), pr.data_ptr(),
                                       x.numel(), C.byref(p), seeds.data_ptr(), L, 0, ctr.data_ptr(), 0, eps2.data_ptr(),
                                       flag.data_ptr(), _lib.stream_of(x))
    _lib.check(rc, "mcmc_update_philox")
    assert torch.equal(eps2, eps) and torch.equal(x2, x)
    # a chain's noise does not depend on its place in the batch
    x3, eps3 = x0[1:2].clone(), torch.empty_like(x[1:2])
    rc = _lib.lib().mcmc_update_philox(x3.data_ptr(), v0[1:2].clone().data_ptr(), g[1:2].contiguous().data_ptr(),
                                       xt[1:2].contiguous().data_ptr(), pr[1:2].contiguous().data_ptr(), L, C.byref(p),
                                       seeds[1:2].contiguous().data_ptr(), L, step, None, 0, eps3.data_ptr(),
                                       flag.data_ptr(), _lib.stream_of(x))
    _lib.check(rc, "mcmc_update_philox")
    assert torch.equal(eps3[0], eps[1]) and torch.equal(x3[0], x[1])
    assert flag.item() == 0


def test_philox_kernel_takes_any_chain_length_and_offset_views(dev):
    """chain_len % 4 != 0 or buffers off 16-byte alignment: the element-at-a-time path draws the
    numbers of the float4 path (element j of a chain = normal j & 3 of call j >> 2), linear and
    quad layout -- MCMCSampler(noise='philox') on latent shapes that are not multiples of 4."""
    from step_amd import _lib
    from oracle.philox_oracle import chain_noise_quad
    p = _lib.mcmc_params_t(0.7, 0.3, 1e-2, 0.1, 0.1414, 0.0, 1.4142, 0)      # langevin
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    gen = torch.Generator(device=dev).manual_seed(2)
    for L, W, off in ((1001, 0, 0), (4 * 250, 0, 1), (6 * 10, 10, 0), (8 * 12, 12, 3)):
        B = 2
        seeds = torch.tensor([5, 2 ** 33 + 1], dtype=torch.int64, device=dev)
        buf = [torch.randn(B * L + 8, device=dev, generator=gen) for _ in range(4)]
        x, g, xt, eps = (b[off:off + B * L] for b in buf)                     # offset views
        x0 = x.clone()
        rc = _lib.lib().mcmc_update_philox(x.data_ptr(), None, g.data_ptr(), xt.data_ptr(), None, B * L, C.byref(p),
                                           seeds.data_ptr(), L, 11, None, W, eps.data_ptr(), flag.data_ptr(),
                                           _lib.stream_of(x))
        _lib.check(rc, "mcmc_update_philox")
        Lp = (L + 3) // 4 * 4
        if W == 0:
            want = np.stack([chain_noise(int(sd), 11, Lp)[:L] for sd in seeds.tolist()])
        else:
            rows = (L // W + 3) // 4 * 4
            want = np.stack([chain_noise_quad(int(sd), 11, rows * W, W)[:L] for sd in seeds.tolist()])
        got = eps.view(B, L).cpu().numpy()
        assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max(), (L, W, off)
        f = lambda v: torch.tensor(v, dtype=torch.float32).item()
        s = -g * f(0.7) + (xt - x0) * f(0.3)
        assert (x - (x0 + f(1e-2) * s + f(0.1414) * eps)).abs().max().item() <= 1e-5


DAPS = dict(ann=dict(name="edm", num_steps=4, sigma_max=10.0, sigma_min=0.1),
            dif=dict(name="edm", num_steps=3, sigma_min=0.01),
            mc=dict(num_steps=5, lr=2e-6, tau=0.05, lr_min_ratio=0.1, prior_solver="gaussian",
                    mc_algo="hmc", momentum=0.5))


@pytest.mark.parametrize("noise", ["torch", "philox"])
@pytest.mark.parametrize("latent", [True, False])
def test_sample_posterior_is_invariant_to_batching(noise, latent, dev):
    """sample_posterior with LatentDAPS and the HIP black-hole operator (64 x 64, 8 chains): every
    chain's sample is the same whether the chains run as one batch of 8, in chunks of 3, or alone
    (reference loop sample.py:64-65 runs them one by one)."""
    from step_amd import driver, eht_lite
    from step_amd.forward_operator import BlackHoleImagingTorch
    from step_amd.sampler import LatentDAPS
    tab = eht_lite.synthetic_table(5, 64, seed=3, sigma_scale=5e-3)
    op = BlackHoleImagingTorch(imsize=64, ref_multiplier=0.001, device=dev, obs_table=tab, w_flux=0.26)
    model = ToyModel((64, 1, 32, 32), factor=2, gain=0.05).to(dev)
    y = op(torch.rand(1, 64, 1, 64, 64, device=dev) * 2 - 1)
    s = LatentDAPS(DAPS["ann"], DAPS["dif"], dict(DAPS["mc"], noise=noise), latent_wrapper=latent)
    runs = [driver.sample_posterior(s, model, op, y, 8, seed=37, device=dev, max_batch=mb) for mb in (None, 3, 1)]
    assert runs[0].shape == (8, 64, 1, 64, 64) and torch.isfinite(runs[0]).all()
    assert (runs[0][0] - runs[0][1]).abs().max() > 1e-3        # chains differ
    for r in runs[1:]:
        for c in range(8):
            assert_close(r[c], runs[0][c], 2e-5, "chain %d" % c)
    again = driver.sample_posterior(s, model, op, y, 8, seed=37, device=dev)
    assert torch.equal(again, runs[0])                         # and the run is reproducible
    other = driver.sample_posterior(s, model, op, y, 8, seed=37, device=dev, item_offset=1000)
    assert (other - runs[0]).abs().max() > 1e-3                # another video: other noise


def test_per_chain_nan_guard(dev):
    """nan_guard='chain' zeroes only the chain that went NaN; the default zeroes the batch
    (daps.py:202-203)."""
    from step_amd.forward_operator import get_operator
    from step_amd.sampler import MCMCSampler
    op = get_operator("mri", acc_factor=6, mode="same_mask", resolution=192, frames=12, num_acs_lines=24)
    gen = torch.Generator(device=dev).manual_seed(2)
    x0 = torch.randn(3, 12, 2, 192, 192, device=dev, generator=gen)
    y = op(torch.randn(1, 12, 2, 192, 192, device=dev, generator=gen))
    x0[1, 0, 0, 5, 5] = float("nan")
    kw = dict(num_steps=2, lr=1e-6, tau=1.0, mc_algo="langevin", prior_solver="gaussian")
    out = MCMCSampler(**kw).sample(x0.clone(), None, x0.clone(), op, y, 0.5, 0.0)
    assert (out == 0).all()
    out = MCMCSampler(nan_guard="chain", **kw).sample(x0.clone(), None, x0.clone(), op, y, 0.5, 0.0)
    assert (out[1] == 0).all() and torch.isfinite(out).all() and out[0].abs().max() > 0 and out[2].abs().max() > 0
