"""GPU parity of the any-size dynamic-MRI kernels (csrc/mri_any.hip, mri_forward_any / mri_adjoint_any /
mri_loss_grad_any through the C ABI) behind the reference's `resolution` argument
(forward_operator/dynamic_mri.py:9,42-48; loss / gradient base.py:40-49).

Pinning: at 192 x 192 the same kernels are held against the golden fixtures written from the reference
(tests/golden/mri_*.npz) and against the tuned kernels; at 128, 160, 75, 97 and 256 against
tests/golden/mri_sizes.npz, written from the reference's own operator at those resolutions
(oracle/gen_golden_mri_sizes.py); at the other sizes against oracle/mri_oracle.py, the restatement all of
these fixtures pin."""
import numpy as np
import pytest
import torch

from oracle.mri_oracle import MRIOracle
from tests.helpers import RTOL, assert_close, t

pytestmark = pytest.mark.gpu


def make(res, mode="same_mask", acc=4, acs=8, frames=5):
    from step_amd.forward_operator import get_operator
    return (get_operator("mri", acc_factor=acc, mode=mode, resolution=res, frames=frames, num_acs_lines=acs),
            MRIOracle(acc, mode, res, frames, acs))


@pytest.mark.parametrize("mode", ["same_mask", "split_mask", "random_split_mask"])
def test_any_size_kernels_match_golden_at_192(mode, golden, dev):
    from tests.test_gpu_mri import inputs, make_op
    g = golden("mri_" + mode)
    op = make_op(mode, sigma=float(g["noise"]))
    op.generic_kernels = True
    assert not op.has_descent_step
    x_gt, x, nz = inputs(g)
    y = op(x_gt.to(dev)) + float(g["noise"]) * nz.to(dev)
    fwd = op(x.to(dev))
    assert (fwd.cpu().flatten()[::11] - t(g["fwd_sub"])).abs().max().item() <= RTOL * float(np.abs(g["fwd_sub"]).max())
    assert_close(fwd[0, 0], g["fwd_frame0"], RTOL, "forward frame 0")
    grad, lsum = op.gradient(x.to(dev), y, return_loss=True)
    assert (grad.cpu().flatten()[::11] - t(g["grad_sub"])).abs().max().item() <= RTOL * float(np.abs(g["grad_sub"]).max())
    assert_close(grad[0, 0], g["grad_frame0"], RTOL, "gradient frame 0")
    assert_close((grad.double() ** 2).flatten(1).sum(-1), g["grad_sq_sum"], RTOL, "gradient energy")
    assert_close(lsum, g["loss_sum"], RTOL, "loss sum")
    assert_close(op.loss(x.to(dev), y), g["loss"], RTOL, "loss")
    # and against the tuned kernels on the same inputs: two implementations, one answer
    tuned = make_op(mode)
    g2, l2 = tuned.gradient(x.to(dev), y, return_loss=True)
    assert_close(grad, g2, 1e-5, "gradient, any-size vs tuned")
    assert_close(lsum, l2, 1e-5, "loss, any-size vs tuned")
    assert_close(fwd, tuned(x.to(dev)), 1e-5, "forward, any-size vs tuned")


# powers of two, 3- and 5-smooth sizes, a size with the factor 7, an odd size (fftshift by (n - 1) / 2: a true
# phase, not a sign), a prime (one direct pass), the largest size
@pytest.mark.parametrize("res,mode,acc,acs,frames", [(64, "same_mask", 4, 8, 5), (96, "split_mask", 3, 6, 4),
                                                     (128, "random_split_mask", 4, 12, 6), (160, "same_mask", 6, 16, 3),
                                                     (200, "same_mask", 5, 10, 2), (224, "split_mask", 4, 8, 3),
                                                     (256, "same_mask", 8, 24, 3), (320, "random_split_mask", 6, 20, 4),
                                                     (384, "same_mask", 6, 24, 2), (75, "same_mask", 3, 7, 3),
                                                     (97, "split_mask", 2, 5, 4), (8, "same_mask", 2, 2, 2),
                                                     (512, "same_mask", 1, 0, 1), (1024, "same_mask", 16, 32, 1)])
def test_other_resolutions_match_oracle(res, mode, acc, acs, frames, dev):
    op, orc = make(res, mode, acc, acs, frames)
    assert torch.equal(op.mask, orc.mask)
    gen = torch.Generator().manual_seed(res)
    B = 3 if res <= 256 else 2
    x = torch.randn(B, frames, 2, res, res, generator=gen)
    for yb in (1, B):
        y = orc(torch.randn(yb, frames, 2, res, res, generator=gen)) + 0.1 * torch.randn(yb, frames, 2, res, res, generator=gen)
        g_ref, l_ref = orc.gradient(x, y, return_loss=True)
        g_got, l_got = op.gradient(x.to(dev), y.to(dev), return_loss=True)
        assert_close(l_got, l_ref, RTOL, "loss")
        assert_close(g_got, g_ref, RTOL, "gradient")
        assert_close(op.loss(x.to(dev), y.to(dev)), orc.loss(x, y), RTOL, "per-chain loss")
    assert_close(op(x.to(dev)), orc(x), RTOL, "forward")
    assert not op.has_descent_step and not op.has_langevin_step


@pytest.mark.parametrize("res,width", [(192, 160), (64, 100), (100, 33), (192, 2)])
def test_frames_that_are_not_square(res, width, dev):
    """The reference's mask broadcasts over the last axis (dynamic_mri.py:15,48): any frame width goes."""
    op, orc = make(res, "split_mask", 3, 8, 4)
    gen = torch.Generator().manual_seed(width)
    x = torch.randn(2, 4, 2, res, width, generator=gen)
    y = orc(torch.randn(1, 4, 2, res, width, generator=gen)) + 0.1 * torch.randn(1, 4, 2, res, width, generator=gen)
    assert_close(op(x.to(dev)), orc(x), RTOL, "forward")
    g_ref, l_ref = orc.gradient(x, y, return_loss=True)
    g_got, l_got = op.gradient(x.to(dev), y.to(dev), return_loss=True)
    assert_close(l_got, l_ref, RTOL, "loss")
    assert_close(g_got, g_ref, RTOL, "gradient")


def test_autograd_routes_and_adjoint_identity_at_another_size(dev):
    op, orc = make(128, "same_mask", 4, 12, 3)
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(2, 3, 2, 128, 128, generator=gen)
    y = torch.randn(1, 3, 2, 128, 128, generator=gen)
    w = torch.tensor([0.5, -2.0])
    xr = x.clone().requires_grad_(True)
    (orc.loss(xr, y) * w).sum().backward()
    xd = x.to(dev).requires_grad_(True)
    (op.loss(xd, y.to(dev)) * w.to(dev)).sum().backward()
    assert_close(xd.grad, xr.grad, RTOL, "weighted loss backward")
    xd2 = x.to(dev).requires_grad_(True)
    ((op(xd2) - y.to(dev)) ** 2).flatten(1).sum(-1).sum().backward()
    assert_close(xd2.grad, op.gradient(x.to(dev), y.to(dev)), RTOL, "autograd through op(x)")
    gk = torch.randn(2, 3, 2, 128, 128, generator=gen).to(dev)
    lhs = (op(x.to(dev)) * gk).sum().double()
    rhs = (x.to(dev) * op._launch_adjoint(gk)).sum().double()
    assert abs(lhs - rhs) / abs(lhs) <= 1e-4


def test_ragged_and_empty_masks_at_another_size(dev):
    """One kept row in all (every other mask frame empty: loss = |y|^2, gradient 0) and every row kept."""
    for acc, mode in ((128, "split_mask"), (1, "same_mask")):
        op, orc = make(128, mode, acc, 0, 4)
        gen = torch.Generator().manual_seed(acc)
        x = torch.randn(2, 4, 2, 128, 128, generator=gen)
        y = orc(torch.randn(2, 4, 2, 128, 128, generator=gen)) + 0.1 * torch.randn(2, 4, 2, 128, 128, generator=gen)
        g_ref, l_ref = orc.gradient(x, y, return_loss=True)
        g_got, l_got = op.gradient(x.to(dev), y.to(dev), return_loss=True)
        assert_close(l_got, l_ref, RTOL, "loss")
        assert_close(g_got, g_ref, RTOL, "gradient")
        assert_close(op(x.to(dev)), orc(x), RTOL, "forward")


def test_sampler_runs_at_another_size(dev):
    """MCMCSampler.sample / optimize on a 128 x 128 operator: the gradient + update-kernel route (the in-kernel
    updates are the 192 x 192 kernels')."""
    from step_amd.sampler.daps import MCMCSampler
    op, orc = make(128, "same_mask", 4, 12, 3)
    gen = torch.Generator().manual_seed(3)
    truth = torch.randn(1, 3, 2, 128, 128, generator=gen)
    y = orc(truth).to(dev)
    x0hat = (truth + 0.3 * torch.randn(4, 3, 2, 128, 128, generator=gen)).to(dev)
    s = MCMCSampler(num_steps=8, lr=2e-5, tau=0.1, lr_min_ratio=0.1, prior_solver="gaussian", mc_algo="langevin")
    xt = x0hat + 0.5 * torch.randn(x0hat.shape, generator=gen).to(dev)
    out = s.sample(xt, None, x0hat, op, y, 1.0, 0.5)
    assert torch.isfinite(out).all() and out.shape == x0hat.shape
    opt = MCMCSampler(num_steps=8, lr=2e-5, tau=0.1).optimize(x0hat, op, y, 1.0, 0.5)
    assert torch.isfinite(opt).all()
    assert op.loss(opt, y).sum().item() < op.loss(x0hat, y).sum().item()


def test_refusals(dev):
    from step_amd import _lib
    assert _lib.lib().mri_any_ws_bytes(1, 1, 1025, 64, 4) == 0
    assert _lib.lib().mri_any_ws_bytes(1, 1, 64, 1, 4) == 0
    op, _ = make(64)
    x = torch.zeros(1, 5, 2, 64, 64, device=dev)
    rows, nrows = op._tables(x.device)
    ws = torch.empty(16, device=dev, dtype=torch.uint8)
    rc = _lib.lib().mri_forward_any(x.data_ptr(), 1, 5, 64, 64, rows.data_ptr(), nrows.data_ptr(), 1, op._max_rows,
                                    ws.data_ptr(), ws.numel(), torch.empty_like(x).data_ptr(), _lib.stream_of(x))
    assert rc != 0 and "workspace" in _lib.lib().step_last_error().decode()
    with pytest.raises(ValueError, match="expects"):
        op(torch.zeros(1, 5, 2, 32, 64, device=dev))          # the mask is 64 rows high


@pytest.mark.parametrize("res", [1021, 1023, 998, 6, 3])
def test_awkward_sizes(res, dev):
    """A prime near the limit (one direct pass and a second buffer, few lines per workgroup), 3 x 11 x 31,
    2 x 499, and frames smaller than a strip."""
    frames = 1 if res > 100 else 3
    op, orc = make(res, "same_mask", 4, min(8, res // 2), frames)
    gen = torch.Generator().manual_seed(res)
    x = torch.randn(2, frames, 2, res, res, generator=gen)
    y = orc(torch.randn(1, frames, 2, res, res, generator=gen)) + 0.1 * torch.randn(1, frames, 2, res, res, generator=gen)
    g_ref, l_ref = orc.gradient(x, y, return_loss=True)
    g_got, l_got = op.gradient(x.to(dev), y.to(dev), return_loss=True)
    assert_close(l_got, l_ref, RTOL, "loss")
    assert_close(g_got, g_ref, RTOL, "gradient")
    assert_close(op(x.to(dev)), orc(x), RTOL, "forward")


def test_other_resolutions_match_the_reference_fixtures(golden, dev):
    """tests/golden/mri_sizes.npz: outputs of the REFERENCE's DynamicMRI at 128, 160, 75 (odd), 97 (prime) and 256,
    written by oracle/gen_golden_mri_sizes.py -- the any-size kernels against the reference itself, not only
    against the oracle."""
    from step_amd.forward_operator import get_operator
    from tests.test_oracle_golden import mri_size_cases, mri_size_inputs
    g = golden("mri_sizes")
    for i, res, mode, acc, acs, fr, B, noise in mri_size_cases(g):
        op = get_operator("mri", acc_factor=acc, mode=mode, resolution=res, frames=fr, num_acs_lines=acs, sigma=noise)
        k = "c%d_" % i
        assert np.array_equal(op.mask[0, :, 0, :, 0].numpy().astype(np.uint8), g[k + "mask"])
        x_gt, x, nz = mri_size_inputs(res, fr, B)
        y = op(x_gt.to(dev)) + noise * nz.to(dev)
        fwd = op(x.to(dev))
        assert (fwd.cpu().flatten()[::7] - t(g[k + "fwd_sub"])).abs().max().item() <= RTOL * float(np.abs(g[k + "fwd_sub"]).max())
        assert_close(fwd[0, 0], g[k + "fwd_frame0"], RTOL, "forward frame 0 at %d" % res)
        grad, lsum = op.gradient(x.to(dev), y, return_loss=True)
        assert (grad.cpu().flatten()[::7] - t(g[k + "grad_sub"])).abs().max().item() <= RTOL * float(np.abs(g[k + "grad_sub"]).max())
        assert_close(grad[0, 0], g[k + "grad_frame0"], RTOL, "gradient frame 0 at %d" % res)
        assert_close((grad.double() ** 2).flatten(1).sum(-1), g[k + "grad_sq_sum"], RTOL, "gradient energy at %d" % res)
        assert_close(lsum, g[k + "loss_sum"], RTOL, "loss sum at %d" % res)
        assert_close(op.loss(x.to(dev), y), g[k + "loss"], RTOL, "loss at %d" % res)
