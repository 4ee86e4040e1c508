"""GPU parity tests of the fused dynamic-MRI kernels (through the C ABI) against the golden
fixtures generated from the reference and against the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle.mri_oracle import MRIOracle
from tests.helpers import RTOL, assert_close, t

pytestmark = pytest.mark.gpu


def inputs(g):
    rng = np.random.RandomState(int(g["seed"]))
    B = int(g["B"])
    shape = (B, 12, 2, 192, 192)
    x_gt = torch.from_numpy(rng.randn(*shape).astype(np.float32))
    x = torch.from_numpy(rng.randn(*shape).astype(np.float32))
    nz = torch.from_numpy(rng.randn(*shape).astype(np.float32))
    return x_gt, x, nz


def make_op(mode, **kw):
    from step_amd.forward_operator import get_operator
    cfg = dict(acc_factor=6, mode=mode, resolution=192, frames=12, num_acs_lines=24)
    cfg.update(kw)
    return get_operator("mri", **cfg)


@pytest.mark.parametrize("mode", ["same_mask", "split_mask", "random_split_mask"])
def test_matches_golden(mode, golden, dev):
    g = golden("mri_" + mode)
    op = make_op(mode, sigma=float(g["noise"]))
    assert np.array_equal(op.mask[0, :, 0, :, 0].numpy().astype(np.uint8), g["mask"])
    x_gt, x, nz = inputs(g)
    y = op(x_gt.to(dev)) + float(g["noise"]) * nz.to(dev)
    fwd = op(x.to(dev))
    scale = float(np.abs(g["fwd_sub"]).max())
    assert (fwd.cpu().flatten()[::11] - t(g["fwd_sub"])).abs().max().item() <= RTOL * scale
    assert_close(fwd[0, 0], g["fwd_frame0"], RTOL, "forward frame 0")
    assert_close((fwd.double() ** 2).flatten(1).sum(-1), g["fwd_sq_sum"], RTOL, "forward energy")
    grad, lsum = op.gradient(x.to(dev), y, return_loss=True)
    gscale = float(np.abs(g["grad_sub"]).max())
    assert (grad.cpu().flatten()[::11] - t(g["grad_sub"])).abs().max().item() <= RTOL * gscale
    assert_close(grad[0, 0], g["grad_frame0"], RTOL, "gradient frame 0")
    assert_close((grad.double() ** 2).flatten(1).sum(-1), g["grad_sq_sum"], RTOL, "gradient energy")
    assert_close(lsum, g["loss_sum"], RTOL, "loss sum")
    assert_close(op.loss(x.to(dev), y), g["loss"], RTOL, "loss")


@pytest.mark.parametrize("mode,acc,acs", [("same_mask", 6, 24), ("split_mask", 8, 12),
                                          ("random_split_mask", 6, 24), ("same_mask", 2, 24),
                                          ("same_mask", 1, 0), ("split_mask", 1, 0)])
def test_matches_oracle(mode, acc, acs, dev):
    """Includes masks with more kept rows than the LDS row buffer (acc 2: 108 rows, acc 1: all
    192) which run in row groups with gradient accumulation."""
    op = make_op(mode, acc_factor=acc, num_acs_lines=acs)
    orc = MRIOracle(acc, mode, 192, 12, acs)
    assert torch.equal(op.mask, orc.mask)
    gen = torch.Generator().manual_seed(acc * 100 + acs)
    x = torch.randn(2, 12, 2, 192, 192, generator=gen)
    y = orc(torch.randn(2, 12, 2, 192, 192, generator=gen)) \
        + 0.1 * torch.randn(2, 12, 2, 192, 192, generator=gen)
    assert_close(op(x.to(dev)), orc(x), RTOL, "forward")
    g_ref, l_ref = orc.gradient(x, y, return_loss=True)
    g_got, l_got = op.gradient(x.to(dev), y.to(dev), return_loss=True)
    assert_close(l_got, l_ref, RTOL, "loss")
    assert_close(g_got, g_ref, RTOL, "gradient")
    assert_close(op.loss(x.to(dev), y.to(dev)), orc.loss(x, y), RTOL, "per-chain loss")


def test_autograd_routes(dev):
    """loss() and op(x) inside an autograd graph (LatentWrapper / custom callers) agree with the
    fused gradient() and with the oracle's autograd."""
    op, orc = make_op("same_mask"), MRIOracle(6, "same_mask", 192, 12, 24)
    gen = torch.Generator().manual_seed(7)
    x = torch.randn(2, 12, 2, 192, 192, generator=gen)
    y = torch.randn(1, 12, 2, 192, 192, generator=gen)       # batch-1 measurement broadcasts
    w = torch.tensor([0.5, -2.0])
    xr = x.clone().requires_grad_(True)
    (orc.loss(xr, y) * w).sum().backward()
    xd = x.to(dev).requires_grad_(True)
    (op.loss(xd, y.to(dev)) * w.to(dev)).sum().backward()
    assert_close(xd.grad, xr.grad, RTOL, "weighted loss backward")
    # through op(x): generic autograd of sum((A x - y)^2) like base.py:40-41
    xd2 = x.to(dev).requires_grad_(True)
    ((op(xd2) - y.to(dev)) ** 2).flatten(1).sum(-1).sum().backward()
    g_fused = op.gradient(x.to(dev), y.to(dev))
    assert_close(xd2.grad, g_fused, RTOL, "autograd through op(x)")
    # adjoint identity <A x, g> = <x, A^H g>
    gk = torch.randn(2, 12, 2, 192, 192, generator=gen).to(dev)
    lhs = (op(x.to(dev)) * gk).sum().double()
    rhs = (x.to(dev) * op._launch_adjoint(gk)).sum().double()
    assert abs(lhs - rhs) / abs(lhs) <= 1e-4


def test_config_batch_properties(dev):
    """BASELINE config size (B = 32): chains are independent and the forward is linear."""
    op = make_op("same_mask")
    gen = torch.Generator(device=dev).manual_seed(2)
    x1 = torch.randn(32, 12, 2, 192, 192, device=dev, generator=gen)
    x2 = torch.randn(32, 12, 2, 192, 192, device=dev, generator=gen)
    y = op(x2[:1])
    assert_close(op(1.5 * x1 - x2), 1.5 * op(x1) - op(x2), RTOL, "linearity")
    g, _ = op.gradient(x1, y, return_loss=True)
    g5 = op.gradient(x1[5:6], y)
    assert torch.equal(g[5:6], g5)
    # zero residual at the truth: loss 0, gradient 0
    g0, l0 = op.gradient(x2[:1], y, return_loss=True)
    assert l0.item() <= 1e-6 * (y ** 2).sum().item()
    assert g0.abs().max().item() <= 1e-3 * g.abs().max().item()


def test_tuned_entry_points_refuse_other_resolutions_through_the_abi(dev):
    """The 192 x 192 kernels say so when handed another size (the operator sends those to mri_*_any,
    tests/test_gpu_mri_any.py)."""
    from step_amd import _lib
    x = torch.zeros(1, 5, 2, 64, 64, device=dev)
    rows = torch.zeros(1, 64, dtype=torch.int32, device=dev)
    nrows = torch.ones(1, dtype=torch.int32, device=dev)
    rc = _lib.lib().mri_forward(x.data_ptr(), 1, 5, 64, rows.data_ptr(), nrows.data_ptr(), 1, 1,
                                torch.empty_like(x).data_ptr(), _lib.stream_of(x))
    assert rc != 0 and "only resolution 192" in _lib.lib().step_last_error().decode()


@pytest.mark.parametrize("mode,acc,acs,frames", [("split_mask", 192, 0, 12),      # one kept row in all: 11 empty frames
                                                 ("random_split_mask", 48, 0, 12),  # 4 rows over 12 frames
                                                 ("same_mask", 3, 0, 12),          # 64 rows: a full row buffer
                                                 ("same_mask", 4, 16, 5)])          # another frame count
def test_ragged_and_empty_masks(mode, acc, acs, frames, dev):
    """Mask frames with no kept row at all (loss = |y|^2, gradient 0), with a single row, and with
    exactly as many rows as the row buffer holds; a shared and a per-chain measurement."""
    from step_amd.forward_operator import get_operator
    op = get_operator("mri", acc_factor=acc, mode=mode, resolution=192, frames=frames, num_acs_lines=acs)
    orc = MRIOracle(acc, mode, 192, frames, acs)
    assert torch.equal(op.mask, orc.mask)
    gen = torch.Generator().manual_seed(acc + frames)
    x = torch.randn(3, frames, 2, 192, 192, generator=gen)
    for yb in (1, 3):
        y = orc(torch.randn(yb, frames, 2, 192, 192, generator=gen)) \
            + 0.1 * torch.randn(yb, frames, 2, 192, 192, generator=gen)
        g_ref, l_ref = orc.gradient(x, y, return_loss=True)
        g_got, l_got = op.gradient(x.to(dev), y.to(dev), return_loss=True)
        assert_close(l_got, l_ref, RTOL, "loss")
        assert_close(g_got, g_ref, RTOL, "gradient")
        assert_close(op.loss(x.to(dev), y.to(dev)), orc.loss(x, y), RTOL, "per-chain loss")
    assert_close(op(x.to(dev)), orc(x), RTOL, "forward")


def test_fused_step_equals_forward_then_adjoint(dev):
    """The pipelined loss+gradient kernel against the separate forward and adjoint kernels at the
    bench size: grad = 2 A^H (A x - y), loss = |A x - y|^2, 64 chains, one measurement."""
    op = make_op("same_mask")
    gen = torch.Generator(device=dev).manual_seed(11)
    x = torch.randn(64, 12, 2, 192, 192, device=dev, generator=gen)
    y = op(torch.randn(1, 12, 2, 192, 192, device=dev, generator=gen)) \
        + 0.05 * torch.randn(1, 12, 2, 192, 192, device=dev, generator=gen)
    g, lsum = op.gradient(x, y, return_loss=True)
    r = op(x) - y
    assert_close(g, 2.0 * op._launch_adjoint(r.contiguous()), RTOL, "gradient vs A^H residual")
    assert_close(op.loss(x, y), (r.double() ** 2).flatten(1).sum(-1), RTOL, "loss vs forward")
    assert_close(lsum, (r.double() ** 2).sum(), RTOL, "loss sum")


@pytest.mark.parametrize("y_per_chain", [False, True])
def test_bench_batch_256_against_the_oracle_on_a_subsample(y_per_chain, dev):
    """B = 256 (bench.py's MRI batch: 12 rounds of workgroups on 256 CUs) is checked against the
    oracle on a sub-sample of the chains -- first, last and a few in between, so that every round of
    the grid and both measurement layouts (one for all chains / one per chain) are covered."""
    B, pick = 256, [0, 1, 15, 16, 127, 200, 255]
    op, orc = make_op("same_mask"), MRIOracle(6, "same_mask", 192, 12, 24)
    gen = torch.Generator(device=dev).manual_seed(11)
    x = torch.randn(B, 12, 2, 192, 192, device=dev, generator=gen)
    y = op(torch.randn(B if y_per_chain else 1, 12, 2, 192, 192, device=dev, generator=gen)) \
        + 0.1 * torch.randn(B if y_per_chain else 1, 12, 2, 192, 192, device=dev, generator=gen)
    grad, lsum = op.gradient(x, y, return_loss=True)
    loss = op.loss(x, y)
    xs = x[pick].cpu()
    ys = y[pick].cpu() if y_per_chain else y.cpu()
    g_ref = orc.gradient(xs, ys)
    l_ref = orc.loss(xs, ys)
    assert_close(loss[pick], l_ref, RTOL, "per-chain loss")
    for i, b in enumerate(pick):
        assert_close(grad[b], g_ref[i], RTOL, "gradient of chain %d" % b)
    assert_close(lsum, loss.double().sum(), RTOL, "loss total")
    assert torch.isfinite(grad).all()


@pytest.mark.parametrize("mode,acc,acs,B", [("same_mask", 6, 24, 17), ("same_mask", 6, 24, 50),
                                            ("split_mask", 8, 12, 45), ("random_split_mask", 6, 24, 70)])
def test_persistent_kernel_is_bit_identical_to_the_pipelined_one(mode, acc, acs, B, dev):
    """variant 4 (mri_lg2p.h: a workgroup walks several frames, the next frame's first strips
    travel under the current frame's row and inverse phases) runs the arithmetic of variant 2 in the
    same order: identical bits, with one frame per workgroup (B*T < CUs), several, and ragged."""
    op = make_op(mode, acc_factor=acc, num_acs_lines=acs)
    gen = torch.Generator().manual_seed(17 * acc + B)
    x = torch.randn(B, 12, 2, 192, 192, generator=gen).to(dev)
    y = (op(torch.randn(1, 12, 2, 192, 192, generator=gen).to(dev))
         + 0.1 * torch.randn(1, 12, 2, 192, 192, generator=gen).to(dev))
    w = torch.linspace(0.5, 1.5, B, device=dev)
    res = {}
    for v in (2, 4):
        op.kernel_variant = v
        g, l = op.gradient(x, y, return_loss=True)
        res[v] = (g, l, op.loss(x, y))
    assert torch.equal(res[4][0], res[2][0])
    assert torch.equal(res[4][2], res[2][2])
    assert torch.equal(res[4][1], res[2][1])


@pytest.mark.parametrize("mode,acc,acs,B,parts", [
    ("same_mask", 6, 24, 1, 6), ("same_mask", 6, 24, 1, 3), ("same_mask", 6, 24, 1, 2),
    ("same_mask", 6, 24, 3, 6), ("same_mask", 6, 24, 4, 3), ("same_mask", 6, 24, 10, 2),
    ("split_mask", 8, 12, 2, 6), ("random_split_mask", 6, 24, 5, 3), ("same_mask", 12, 0, 2, 6)])
def test_split_frame_kernel_matches_the_unsplit_one(mode, acc, acs, B, parts, dev):
    """mri_loss_grad_split (mri_lg2s.h: 6 / 3 / 2 workgroups share a frame at batches that leave the
    chip idle) runs the transforms of the pipelined kernel on the same values in the same order:
    the gradient is bit-identical, the loss equal to its summation order; also against the oracle,
    with a chain scale, loss-only, and on two calls in a row through one workspace (the counters
    at its head must come back to zero)."""
    op = make_op(mode, acc_factor=acc, num_acs_lines=acs)
    orc = MRIOracle(acc, mode, 192, 12, acs)
    gen = torch.Generator().manual_seed(31 * acc + 7 * B + parts)
    x = torch.randn(B, 12, 2, 192, 192, generator=gen)
    y = orc(torch.randn(1, 12, 2, 192, 192, generator=gen)) + 0.1 * torch.randn(1, 12, 2, 192, 192, generator=gen)
    xd, yd = x.to(dev), y.to(dev)
    op.split_frames = False
    g0, l0 = op.gradient(xd, yd, return_loss=True)
    loss0 = op.loss(xd, yd)
    op.split_frames = parts
    for rep in range(2):
        g1, l1 = op.gradient(xd, yd, return_loss=True)
        assert torch.equal(g1, g0), "gradient bits (call %d)" % rep
        assert_close(l1, l0, 1e-6, "loss sum")
    loss1 = op.loss(xd, yd)                                   # loss only: no gradient, one wait
    assert_close(loss1, loss0, 1e-6, "per-chain loss")
    ws = op._dev_tables[("split_ws", str(xd.device))]
    assert int(ws[:4 * (2 * B * 12 + 2)].view(torch.int32).abs().sum()) == 0, "counters / error word not zero"
    g_ref, l_ref = orc.gradient(x, y.expand(B, -1, -1, -1, -1), return_loss=True)
    assert_close(g1, g_ref, RTOL, "gradient vs the oracle")
    assert_close(l1, l_ref, RTOL, "loss vs the oracle")
    # auto: picks a split for this batch (or the unsplit kernel from 7 chains on), same bits
    op.split_frames = None
    g2, _ = op.gradient(xd, yd, return_loss=True)
    assert torch.equal(g2, g0)
    # through autograd with a per-chain upstream gradient
    xr = xd.clone().requires_grad_(True)
    wts = torch.linspace(0.5, 1.5, B, device=dev)
    (op.loss(xr, yd) * wts).sum().backward()
    assert_close(xr.grad, g0 * wts.view(-1, 1, 1, 1, 1), 1e-6, "weighted gradient")


def test_split_frame_applies_to_small_batches_only(dev):
    from step_amd import _lib
    L = _lib.lib()
    assert L.mri_loss_grad_split_ws_bytes(1, 12, 52, 0) > 0
    assert L.mri_loss_grad_split_ws_bytes(6, 12, 52, 0) > 0
    assert L.mri_loss_grad_split_ws_bytes(7, 12, 52, 0) == 0 and L.mri_loss_grad_split_ws_bytes(7, 12, 52, 2) > 0          # 88 frames x 3 > 256 CUs
    assert L.mri_loss_grad_split_ws_bytes(4, 12, 108, 0) == 0          # more kept rows than the row buffer
    op = make_op("same_mask")
    x = torch.randn(12, 12, 2, 192, 192, device=dev)
    y = op(torch.randn(1, 12, 2, 192, 192, device=dev))
    op.gradient(x, y)                                               # 12 chains: the unsplit kernel
    assert ("split_ws", str(x.device)) not in op._dev_tables


@pytest.mark.parametrize("mode,acc,acs,B", [("same_mask", 6, 24, 3), ("same_mask", 6, 24, 30), ("split_mask", 8, 12, 5)])
def test_step_with_the_update_inside_equals_gradient_then_update(mode, acc, acs, B, dev):
    """mri_step_update: MCMCSampler.optimize's update and the Langevin / momentum-free HMC step of
    MCMCSampler.sample (noise from a tensor) applied by the inverse column phase where it would have
    stored the gradient == gradient() followed by mcmc_update with xt = x0hat and no prior tensor."""
    import ctypes as C
    from step_amd import _lib
    op = make_op(mode, acc_factor=acc, num_acs_lines=acs)
    gen = torch.Generator(device=dev).manual_seed(41 + B)
    x = torch.randn(B, 12, 2, 192, 192, device=dev, generator=gen)
    x0 = x + 0.05 * torch.randn(x.shape, device=dev, generator=gen)
    eps = torch.randn(x.shape, device=dev, generator=gen)
    y = op(torch.randn(1, 12, 2, 192, 192, device=dev, generator=gen)) + 0.1 * torch.randn(1, 12, 2, 192, 192, device=dev, generator=gen)
    op.split_frames = False
    g, total = op.gradient(x, y, return_loss=True)
    gmax = float(g.abs().max())
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    lr = 1e-4
    for algo in (2, 0, 1):
        p = _lib.mcmc_params_t(30.0 / gmax, 40.0, 0.02 / gmax if algo == 2 else lr, np.sqrt(lr), np.sqrt(2 * lr), 0.0,
                               np.sqrt(2.0), algo)
        want = x.clone()
        vel = torch.zeros_like(x) if algo == 1 else None
        rc = _lib.lib().mcmc_update(want.data_ptr(), _lib.ptr(vel), g.data_ptr(), x0.data_ptr(), None,
                                    None if algo == 2 else eps.data_ptr(), x.numel(), C.byref(p), flag.data_ptr(),
                                    _lib.stream_of(x))
        _lib.check(rc, "mcmc_update")
        got = x.clone()
        if algo == 2:
            got_total = op.descent_step(got, y, x0, p.lr, p.inv_sigma2, return_loss=True)
        else:
            got_total = op.langevin_step(got, y, x0, p, eps=eps, nan_flag=flag, return_loss=True)
        assert_close(got_total, total, 1e-6, "loss at the old x (algo %d)" % algo)
        assert (want - x).abs().max() > 1e-3
        assert_close(got - x, want - x, 2e-5, "update (algo %d)" % algo)
        assert flag.item() == 0
    eps[0, 0, 0, 0, 0] = float("nan")
    op.langevin_step(x.clone(), y, x0, p, eps=eps, nan_flag=flag)
    assert flag.item() == 1
    with pytest.raises(ValueError):
        op.descent_step(got, y, got, 1e-3, 1.0)


def test_mri_sampler_loops_with_and_without_the_update_inside(dev):
    """MCMCSampler.optimize and .sample (torch noise, 'gaussian' solver, hmc at momentum 0) on the MRI
    operator: the routes through descent_step / langevin_step(eps=...) == gradient + update kernel."""
    from step_amd.sampler.daps import MCMCSampler
    op = make_op("same_mask")
    gen = torch.Generator(device=dev).manual_seed(4)
    x0 = torch.randn(8, 12, 2, 192, 192, device=dev, generator=gen)
    xt = x0 + 0.3 * torch.randn(x0.shape, device=dev, generator=gen)
    y = op(torch.randn(1, 12, 2, 192, 192, device=dev, generator=gen))
    tau = 0.1
    lr = float(1e-3 * tau ** 2 / op.gradient(x0, y).abs().max())
    out = {}
    for fuse in (True, False):
        mc = MCMCSampler(num_steps=6, lr=lr, tau=tau, lr_min_ratio=0.1, prior_solver="gaussian", mc_algo="hmc",
                         momentum=0.0)
        mc.fuse_langevin = mc.fuse_descent = fuse
        mc.set_chains(generators=[torch.Generator(device=dev).manual_seed(50 + i) for i in range(8)])
        out[fuse] = (mc.sample(xt, None, x0, op, y, 0.5, 0.25), mc.optimize(x0, op, y, 0.5, 0.25))
    for k, name in ((0, "sample"), (1, "optimize")):
        assert (out[True][k] - x0).abs().max() > 1e-5
        assert_close(out[True][k] - x0, out[False][k] - x0, 1e-4, name)
