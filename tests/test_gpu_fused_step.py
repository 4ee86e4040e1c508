"""The fused two-launch black-hole step (bh_loss_grad_sep3: closure stage inside the DFT kernels)
against the three-kernel path and against the reference fixtures."""
import numpy as np
import pytest
import torch

from tests.helpers import RTOL, assert_close, obs_from_golden, weights_kw

pytestmark = pytest.mark.gpu


def make(dev, n_sites, imsize, seed=5, **kw):
    from step_amd import eht_lite
    from step_amd.forward_operator import BlackHoleImagingTorch
    tab = eht_lite.synthetic_table(n_sites, 64, seed=seed, sigma_scale=kw.pop("sigma_scale", 5e-3))
    cfg = dict(imsize=imsize, ref_multiplier=0.001, device=dev, obs_table=tab, w_flux=0.26)
    cfg.update(kw)
    return BlackHoleImagingTorch(**cfg)


@pytest.mark.parametrize("n_sites,imsize,B", [(8, 256, 1), (8, 256, 5), (8, 256, 70), (5, 64, 3), (7, 128, 9), (4, 64, 300)])
@pytest.mark.parametrize("normalize", [True, False])
def test_fused_equals_three_kernel_path(n_sites, imsize, B, normalize, dev):
    op = make(dev, n_sites, imsize, loss_normalize=normalize)
    assert op.path == "sep3" and op._frame_tab is not None
    gen = torch.Generator(device=dev).manual_seed(B)
    x = torch.rand(B, 64, 1, imsize, imsize, device=dev, generator=gen) * 2 - 1
    y = op(torch.rand(1, 64, 1, imsize, imsize, device=dev, generator=gen) * 2 - 1)
    assert op._fused_ok(y)
    op.fused_max_batch = 10 ** 9                             # the fused kernels at every batch size
    g1, l1 = op.gradient(x, y, return_loss=True)
    chi_f, loss_f = op.last_chi2.clone(), op._fused_step(x, y)[0]
    op.fused_step = False
    g0, l0 = op.gradient(x, y, return_loss=True)
    chi_0, loss_0 = op.last_chi2.clone(), op.loss(x, y)
    op.fused_step = True
    assert_close(l1, l0, 1e-5, "total loss")
    assert_close(loss_f, loss_0, 1e-5, "per-chain loss")
    assert_close(chi_f, chi_0, 1e-5, "raw chi^2")
    # same kernels, same g_vis up to the order of the chi^2 sums: far inside the 1e-4 bar
    for b in range(B):
        assert_close(g1[b], g0[b], 2e-5, "gradient of chain %d" % b)
    # autograd route with the unit-upstream hint == fused gradient, bitwise; without it the
    # three-kernel route
    xr = x[: min(B, 4)].clone().requires_grad_(True)
    ga = torch.autograd.grad(op.loss(xr, y, unit_upstream=True).sum(), xr)[0]
    assert torch.equal(ga, op.gradient(x[: min(B, 4)], y))
    w = torch.linspace(0.5, 2.0, xr.shape[0], device=dev)
    gw = torch.autograd.grad((op.loss(xr, y) * w).sum(), xr)[0]
    assert_close(gw, ga * w.view(-1, 1, 1, 1, 1), 2e-5, "weighted upstream")


def test_fused_matches_reference_fixture_at_full_size(golden, dev):
    """bh_full_wc (64 x 256 x 256, the reference's own outputs) through the fused step."""
    from oracle import eht_synth
    from step_amd import eht_lite
    from step_amd.forward_operator import BlackHoleImagingTorch, BlackHoleObservation
    g = golden("bh_full_wc")
    tab = eht_lite.synthetic_table(8, 64, seed=int(g["obs_seed"]), sigma_scale=float(g["sigma_scale"]))
    op = BlackHoleImagingTorch(imsize=256, ref_multiplier=0.001, device=dev, obs_table=tab, **weights_kw(g))
    y = BlackHoleObservation(*obs_from_golden(g, "y_", dev))
    assert op._fused_ok(y)
    x = torch.from_numpy(eht_synth.ring_video(int(g["seed_x"]))).to(dev)
    grad, lsum = op.gradient(x, y, return_loss=True)
    assert_close(lsum, g["x_loss_sum"], RTOL, "loss sum")
    assert_close(op._fused_step(x, y)[0], g["x_loss"], RTOL, "loss")
    assert_close(op.last_vis, g["x_obs_vis"], RTOL, "vis")
    gc = grad.cpu()
    assert_close(gc.flatten()[::97], g["x_grad_sub"], RTOL, "gradient")
    assert_close(gc.flatten(2).sum(-1), g["x_grad_frame_sum"], RTOL, "per-frame gradient sums")


def test_fallbacks(dev):
    """A measurement per chain, visibility / amplitude weights, or closures that span frames use
    the three-kernel path; the C entry point refuses them."""
    import ctypes as C
    from step_amd import _lib
    op = make(dev, 5, 64)
    x = torch.rand(2, 64, 1, 64, 64, device=dev) * 2 - 1
    y2 = op(torch.rand(2, 64, 1, 64, 64, device=dev) * 2 - 1)
    assert not op._fused_ok(y2)
    op.gradient(x, y2)                                        # runs (three kernels)
    assert not make(dev, 5, 64, w_vis=0.5)._fused_ok(op(x[:1]))
    # a triangle that spans two frames: the host table builder refuses
    cpi = op.cp_index.cpu().numpy().copy()
    cpi[0, 1] += op.F
    ft = np.zeros((64, 2, 64, 2), np.int32)
    cpc, cai = op.cp_conjugate.cpu().numpy(), op.camp_index.cpu().numpy()
    rc = _lib.lib().bh_build_frame_closures(_lib.ptr(cpi), _lib.ptr(cpc), cpi.shape[0], _lib.ptr(cai),
                                            cai.shape[0], 64, op.F, _lib.ptr(ft))
    assert rc == -1 and b"spans frames" in _lib.lib().step_last_error()


@pytest.mark.parametrize("n_sites,imsize,B", [(8, 256, 3), (8, 256, 40), (5, 64, 2), (7, 128, 33)])
def test_descent_step_equals_gradient_then_update(n_sites, imsize, B, dev):
    """descent_step (the update of MCMCSampler.optimize, daps.py:210-223, applied inside the
    adjoint) == gradient() followed by the same arithmetic in torch, on the two-launch path
    (B <= 31) and on the three-kernel path."""
    op = make(dev, n_sites, imsize)
    gen = torch.Generator(device=dev).manual_seed(100 + B)
    x = torch.rand(B, 64, 1, imsize, imsize, device=dev, generator=gen) * 2 - 1
    x0 = x + 0.05 * torch.randn(x.shape, device=dev, generator=gen)
    y = op(torch.rand(1, 64, 1, imsize, imsize, device=dev, generator=gen) * 2 - 1)
    g, total = op.gradient(x, y, return_loss=True)
    # a step size that makes the data term and the prior term comparable in the update
    lr = float(0.02 / g.abs().max())
    inv_s2 = float(0.3 * g.abs().max())
    fp = lambda z: z.double()
    want = (fp(x) - lr * (fp(g) + (fp(x) - fp(x0)) * inv_s2)).float()
    xs = x.clone()
    got_total = op.descent_step(xs, y, x0, lr, inv_s2, return_loss=True)
    assert_close(got_total, total, 1e-6, "loss at the old x")
    # the update relative to the step it takes (x itself is O(1), the step O(0.02))
    assert_close(xs - x, want - x, 2e-5, "update")
    assert (xs - want).abs().max().item() <= 4e-7
    # x0hat given as a broadcastable / non-contiguous tensor
    xs2 = x.clone()
    op.descent_step(xs2, y, x0.transpose(3, 4).contiguous().transpose(3, 4), lr, inv_s2)
    assert torch.equal(xs2, xs)
    with pytest.raises(ValueError):
        op.descent_step(xs2, y, xs2, lr, inv_s2)
    with pytest.raises(ValueError):
        op.descent_step(xs2.transpose(3, 4), y, x0, lr, inv_s2)


def test_optimize_with_and_without_the_fused_descent(dev):
    """MCMCSampler.optimize through descent_step == through gradient + mcmc_update."""
    from step_amd.sampler.daps import MCMCSampler
    op = make(dev, 6, 64)
    gen = torch.Generator(device=dev).manual_seed(3)
    x0 = torch.rand(4, 64, 1, 64, 64, device=dev, generator=gen) * 2 - 1
    y = op(torch.rand(1, 64, 1, 64, 64, device=dev, generator=gen) * 2 - 1)
    g = op.gradient(x0, y)
    mc = MCMCSampler(num_steps=6, lr=float(0.01 / g.abs().max()), tau=0.01)
    a = mc.optimize(x0, op, y, 0.7, 0.2)
    mc.fuse_descent = False
    b = mc.optimize(x0, op, y, 0.7, 0.2)
    assert not torch.equal(a, x0)
    assert_close(a - x0, b - x0, 1e-4, "six descent steps")
