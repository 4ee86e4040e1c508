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


def test_host_frame_table_is_refused_not_dereferenced(dev):
    """bh_build_frame_closures fills a HOST table; the fused step reads it inside a kernel and needs a
    device copy.  A C caller who hands over the host table gets STEP_E_INVALID, not a GPU fault."""
    import ctypes as C
    from step_amd import _lib
    op = make(dev, 5, 64)
    x = torch.rand(2, 64, 1, 64, 64, device=dev) * 2 - 1
    y = op(torch.rand(1, 64, 1, 64, 64, device=dev) * 2 - 1)
    host_tab = op._frame_tab.cpu().numpy().copy()             # pageable host memory
    meas, keep = op._meas(y)
    B, T, M = 2, op.T, op.T * op.F
    buf = torch.zeros(4 * B * T + 2 * B * M + B * T + B * op.cphase_dim + B * op.logcamp_dim + 8 * B, device=dev)
    o = [0, 4 * B * T, 4 * B * T + 2 * B * M, 4 * B * T + 2 * B * M + B * T]
    o.append(o[3] + B * op.cphase_dim)
    o.append(o[4] + B * op.logcamp_dim)
    base = buf.data_ptr()
    ws = _lib.bh_step_ws_t(base + 4 * o[1], base + 4 * o[2], base + 4 * o[3], base + 4 * o[4], base)
    grad = torch.empty(B, T, 64 * 64, device=dev)
    args = lambda tab: (op._ey_t.data_ptr(), op._ey_ri.data_ptr(), op._ex_ri.data_ptr(), T, op.F, 64, 64,
                        x.data_ptr(), B, C.byref(op._tab), tab, C.byref(meas), C.byref(op._weights), C.byref(ws),
                        None, 0.5, base + 4 * o[5], None, None, grad.data_ptr(), None, _lib.stream_of(x))
    rc = _lib.lib().bh_loss_grad_sep3(*args(host_tab.ctypes.data))
    assert rc == -1 and b"DEVICE copy" in _lib.lib().step_last_error()
    rc = _lib.lib().bh_loss_grad_sep3(*args(op._frame_tab.data_ptr()))     # the device copy: runs
    _lib.check(rc, "bh_loss_grad_sep3")
    assert_close(grad.view_as(x), op.gradient(x, y), 1e-6, "raw call vs the operator")


@pytest.mark.parametrize("n_sites,imsize,B", [(8, 256, 1), (8, 256, 3), (8, 256, 40), (5, 64, 2), (7, 128, 1), (7, 128, 33)])
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


@pytest.mark.parametrize("algo", [0, 1])
@pytest.mark.parametrize("n_sites,imsize,B", [(8, 256, 1), (8, 256, 3), (8, 256, 40), (5, 64, 2), (7, 128, 1), (7, 128, 33)])
def test_langevin_step_equals_gradient_then_philox_update(n_sites, imsize, B, algo, dev):
    """langevin_step (the step of MCMCSampler.sample for the 'gaussian' prior solver, applied and
    its noise drawn inside the adjoint) == gradient() followed by mcmc_update_philox in the
    column-quad noise layout with xt = x0hat and no prior tensor, on the two-launch path (B <= 31)
    and on the three-kernel path; the noise itself against the numpy oracle."""
    import ctypes as C
    from oracle.philox_oracle import chain_noise_quad
    from step_amd import _lib
    op = make(dev, n_sites, imsize)
    gen = torch.Generator(device=dev).manual_seed(200 + B)
    x = torch.rand(B, 64, 1, imsize, imsize, device=dev, generator=gen) * 2 - 1
    x0 = x + 0.05 * torch.randn(x.shape, device=dev, generator=gen)
    y = op(torch.rand(1, 64, 1, imsize, imsize, device=dev, generator=gen) * 2 - 1)
    g, total = op.gradient(x, y, return_loss=True)
    gmax = float(g.abs().max())
    lr = 1e-4
    # data term, prior term and noise all of comparable size in the step
    p = _lib.mcmc_params_t(30.0 / gmax, 40.0, lr, np.sqrt(lr), np.sqrt(2 * lr), 0.0, np.sqrt(2.0), algo)
    seeds = torch.arange(B, device=dev, dtype=torch.int64) * 104729 + 2 ** 33 + 1
    step = 2 ** 32 + 3
    L = x[0].numel()
    want, eps = x.clone(), torch.empty_like(x)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    vel = torch.zeros_like(x) if algo == 1 else None
    rc = _lib.lib().mcmc_update_philox(want.data_ptr(), _lib.ptr(vel), g.data_ptr(), x0.data_ptr(), None, x.numel(),
                                       C.byref(p), seeds.data_ptr(), L, step, None, imsize, eps.data_ptr(),
                                       flag.data_ptr(), _lib.stream_of(x))
    _lib.check(rc, "mcmc_update_philox")
    b = B - 1
    ref = chain_noise_quad(int(seeds[b]), step, L, imsize)
    assert np.abs(eps[b].cpu().numpy().ravel() - ref).max() <= 2e-6 * np.abs(ref).max()
    assert abs(float(eps.mean())) < 5e-3 and abs(float(eps.std()) - 1) < 5e-3
    got = x.clone()
    got_total = op.langevin_step(got, y, x0, p, seeds, step=step, nan_flag=flag, return_loss=True)
    assert_close(got_total, total, 1e-6, "loss at the old x")
    assert flag.item() == 0
    assert_close(got - x, want - x, 2e-5, "update")
    assert (got - want).abs().max().item() <= 4e-7
    # the step number from a device word (graph replay); another step gives other noise
    got2 = x.clone()
    ctr = torch.tensor([step], dtype=torch.int64, device=dev)
    op.langevin_step(got2, y, x0, p, seeds, step=0, step_dev=ctr)
    assert torch.equal(got2, got)
    got3 = x.clone()
    op.langevin_step(got3, y, x0, p, seeds, step=step + 1)
    assert (got3 - got).abs().max().item() > 1e-3
    # a chain's step does not depend on the batch it is in
    if B > 1:
        one = x[b:b + 1].clone()
        op.langevin_step(one, y, x0[b:b + 1], p, seeds[b:b + 1].contiguous(), step=step)
        assert_close(one - x[b:b + 1], got[b:b + 1] - x[b:b + 1], 2e-5, "chain alone")
    # NaN guard
    bad = x.clone()
    x0n = x0.clone()
    x0n[0, 3, 0, 5, 7] = float("nan")
    op.langevin_step(bad, y, x0n, p, seeds, step=step, nan_flag=flag)
    assert flag.item() == 1 and torch.isnan(bad).sum().item() == 1
    # what the entry points refuse: a carried velocity, the optimize form, a negative step, bad seeds
    for bad_p in (_lib.mcmc_params_t(1.0, 1.0, lr, np.sqrt(lr), np.sqrt(2 * lr), 0.5, 1.0, 1),
                  _lib.mcmc_params_t(1.0, 1.0, lr, 0.0, 0.0, 0.0, 0.0, 2)):
        with pytest.raises(RuntimeError, match="langevin or hmc with momentum 0"):
            op.langevin_step(x.clone(), y, x0, bad_p, seeds, step=step)
    with pytest.raises(RuntimeError, match="negative step"):
        op.langevin_step(x.clone(), y, x0, p, seeds, step=-1)
    with pytest.raises(ValueError):
        op.langevin_step(x.clone(), y, x0, p, seeds[:1], step=step) if B > 1 else op.langevin_step(x.clone(), y, x0, p, seeds.int(), step=step)
    with pytest.raises(ValueError):
        op.langevin_step(got, y, got, p, seeds, step=step)


@pytest.mark.parametrize("algo", ["langevin", "hmc"])
def test_sample_with_the_update_in_the_adjoint(algo, dev):
    """MCMCSampler.sample(noise='philox', prior_solver='gaussian') through langevin_step equals the
    same loop written out with gradient() + mcmc_update_philox, and switching the route off gives
    a chain with the same statistics but other noise."""
    import ctypes as C
    from step_amd import _lib
    from step_amd.sampler.daps import MCMCSampler
    op = make(dev, 6, 64)
    gen = torch.Generator(device=dev).manual_seed(4)
    B, steps, sigma, ratio = 4, 6, 0.7, 0.2
    x0 = torch.rand(B, 64, 1, 64, 64, device=dev, generator=gen) * 2 - 1
    xt = x0 + sigma * torch.randn(x0.shape, device=dev, generator=gen)
    y = op(torch.rand(1, 64, 1, 64, 64, device=dev, generator=gen) * 2 - 1)
    tau = 0.5
    lr0 = float(1e-3 * tau ** 2 / op.gradient(x0, y).abs().max())
    seeds = torch.arange(B, device=dev, dtype=torch.int64) * 31 + 9
    mc = MCMCSampler(num_steps=steps, lr=lr0, tau=tau, prior_solver="gaussian", mc_algo=algo, momentum=0.0,
                     noise="philox")
    mc.set_chains(seeds=seeds)
    torch.manual_seed(1)
    a = mc.sample(xt, None, x0, op, y, sigma, ratio)
    assert mc.noise_step == steps
    params = mc._params(mc.get_lr(ratio), sigma)
    want = x0.clone()
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    for k in range(steps):
        g = op.gradient(want, y)
        rc = _lib.lib().mcmc_update_philox(want.data_ptr(), None if algo == "langevin" else torch.zeros_like(want).data_ptr(),
                                           g.data_ptr(), x0.data_ptr(), None, want.numel(), C.byref(params),
                                           seeds.data_ptr(), want[0].numel(), k, None, 64, None, flag.data_ptr(),
                                           _lib.stream_of(want))
        _lib.check(rc, "mcmc_update_philox")
    assert not torch.equal(a, x0)
    assert_close(a - x0, want - x0, 1e-4, "six steps")
    mc.fuse_langevin = False
    mc.set_chains(seeds=seeds)
    torch.manual_seed(1)
    b = mc.sample(xt, None, x0, op, y, sigma, ratio)
    assert not torch.equal(a, b)
    da, db = (a - x0).flatten(1), (b - x0).flatten(1)
    assert_close(da.std(dim=1), db.std(dim=1), 2e-2, "step size of the two routes")


@pytest.mark.parametrize("algo", [0, 1])
@pytest.mark.parametrize("n_sites,imsize,B", [(8, 256, 1), (8, 256, 3), (8, 256, 40), (5, 64, 2), (7, 128, 33)])
def test_langevin_step_with_a_noise_tensor_equals_gradient_then_update(n_sites, imsize, B, algo, dev):
    """langevin_step(eps=...): the step of MCMCSampler.sample with the reference's own draw
    (torch.randn_like, daps.py:192) handed to the adjoint as a tensor == gradient() followed by
    mcmc_update with xt = x0hat and no prior tensor, on the two-launch and the three-kernel path."""
    import ctypes as C
    from step_amd import _lib
    op = make(dev, n_sites, imsize)
    gen = torch.Generator(device=dev).manual_seed(300 + B)
    x = torch.rand(B, 64, 1, imsize, imsize, device=dev, generator=gen) * 2 - 1
    x0 = x + 0.05 * torch.randn(x.shape, device=dev, generator=gen)
    eps = torch.randn(x.shape, device=dev, generator=gen)
    y = op(torch.rand(1, 64, 1, imsize, imsize, device=dev, generator=gen) * 2 - 1)
    g, total = op.gradient(x, y, return_loss=True)
    lr = 1e-4
    p = _lib.mcmc_params_t(30.0 / float(g.abs().max()), 40.0, lr, np.sqrt(lr), np.sqrt(2 * lr), 0.0, np.sqrt(2.0), algo)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    want = x.clone()
    vel = torch.zeros_like(x) if algo == 1 else None
    rc = _lib.lib().mcmc_update(want.data_ptr(), _lib.ptr(vel), g.data_ptr(), x0.data_ptr(), None, eps.data_ptr(),
                                x.numel(), C.byref(p), flag.data_ptr(), _lib.stream_of(x))
    _lib.check(rc, "mcmc_update")
    got, eps_before = x.clone(), eps.clone()
    got_total = op.langevin_step(got, y, x0, p, None, eps=eps, nan_flag=flag, return_loss=True)
    assert_close(got_total, total, 1e-6, "loss at the old x")
    assert flag.item() == 0 and torch.equal(eps, eps_before)
    assert_close(got - x, want - x, 2e-5, "update")
    assert (got - want).abs().max().item() <= 4e-7
    # a NaN in the noise raises the flag
    eps[0, 0, 0, 0, 0] = float("nan")
    op.langevin_step(x.clone(), y, x0, p, None, eps=eps, nan_flag=flag)
    assert flag.item() == 1


def test_sample_with_torch_noise_through_the_adjoint(dev):
    """MCMCSampler.sample, torch noise, 'gaussian' solver, hmc at momentum 0 (the shipped black-hole
    algorithm): the route that hands each step's draw to the adjoint == gradient + update kernel on
    the same generator stream."""
    from step_amd.sampler.daps import MCMCSampler
    op = make(dev, 6, 64)
    gen = torch.Generator(device=dev).manual_seed(4)
    x0 = torch.rand(5, 64, 1, 64, 64, device=dev, generator=gen) * 2 - 1
    xt = x0 + 0.3 * torch.randn(x0.shape, device=dev, generator=gen)
    y = op(torch.rand(1, 64, 1, 64, 64, device=dev, generator=gen) * 2 - 1)
    tau = 0.05
    lr = float(1e-3 * tau ** 2 / op.gradient(x0, y).abs().max())
    out = {}
    for fuse in (True, False):
        mc = MCMCSampler(num_steps=8, lr=lr, tau=tau, lr_min_ratio=0.1, prior_solver="gaussian", mc_algo="hmc",
                         momentum=0.0)
        mc.fuse_langevin = fuse
        mc.set_chains(generators=[torch.Generator(device=dev).manual_seed(50 + i) for i in range(5)])
        out[fuse] = mc.sample(xt, None, x0, op, y, 0.5, 0.25)
    assert (out[True] - x0).abs().max() > 1e-4
    assert_close(out[True] - x0, out[False] - x0, 1e-4, "eight steps")
