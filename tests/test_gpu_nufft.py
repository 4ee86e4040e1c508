"""The gridding (NUFFT) path for radial sampling against the direct non-uniform DFT in fp64.
NO REFERENCE COUNTERPART -- parity unpinned; the tolerance is the north_star's 1e-4 relative
(max-norm), which the Kaiser-Bessel design (width 6, oversampling 2) meets with a margin."""
import numpy as np
import pytest
import torch

from oracle.nufft_oracle import NUDFTOracle
from tests.helpers import RTOL, assert_close

pytestmark = pytest.mark.gpu


def make(**kw):
    from step_amd.forward_operator import get_operator
    return get_operator("mri_radial", **kw)


@pytest.mark.parametrize("grid_fft", ["fused", "library"])
@pytest.mark.parametrize("spokes,frames,shared", [(32, 12, False), (13, 3, True)])
def test_matches_direct_nudft(spokes, frames, shared, grid_fft, dev):
    from step_amd.forward_operator import golden_angle_radial
    traj = golden_angle_radial(1 if shared else frames, spokes, 192)
    if shared:                                              # an arbitrary (non-radial) set as well
        rng = np.random.RandomState(0)
        traj = np.concatenate([traj, rng.uniform(-96, 96, (1, 500, 2))], axis=1)
    op, orc = make(spokes=spokes, frames=frames, trajectory=traj), NUDFTOracle(traj, 192)
    op.grid_fft = grid_fft              # nufft_fft.hip's pruned passes / pad + torch.fft + crop
    gen = torch.Generator().manual_seed(spokes)
    x = torch.randn(2, frames, 2, 192, 192, generator=gen)
    y = orc(torch.randn(2, frames, 2, 192, 192, generator=gen)).float() + 0.1 * torch.randn(2, frames, op.J, 2, generator=gen)
    assert_close(op(x.to(dev)), orc(x), RTOL, "type 2 (image -> samples)")
    v = torch.randn(2, frames, op.J, 2, generator=gen)
    assert_close(op._type1(v.to(dev), 1.0), orc.adjoint(v), RTOL, "type 1 (samples -> image)")
    g_ref, l_ref = orc.gradient(x, y, return_loss=True)
    g_got, l_got = op.gradient(x.to(dev), y.to(dev), return_loss=True)
    assert_close(l_got, l_ref, RTOL, "loss")
    assert_close(g_got, g_ref, RTOL, "gradient")
    assert_close(op.loss(x.to(dev), y.to(dev)), orc.loss(x, y), RTOL, "per-chain loss")
    assert_close(op.gradient(x.to(dev), y[:1].to(dev))[1], orc.gradient(x[1:], y[:1])[0], RTOL, "broadcast measurement")


def test_adjoint_identity_and_autograd(dev):
    op = make(spokes=32, frames=12)
    gen = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(3, 12, 2, 192, 192, device=dev, generator=gen)
    v = torch.randn(3, 12, op.J, 2, device=dev, generator=gen)
    lhs = (op(x) * v).flatten(1).sum(-1).double()
    rhs = (x * op._type1(v, 1.0)).flatten(1).sum(-1).double()
    assert ((lhs - rhs).abs() / lhs.abs()).max().item() <= 1e-4          # <A x, v> = <x, A^H v>
    y = op(torch.randn(1, 12, 2, 192, 192, device=dev, generator=gen))
    g = op.gradient(x, y)
    xr = x.clone().requires_grad_(True)
    w = torch.tensor([0.5, -2.0, 1.0], device=dev)
    ga = torch.autograd.grad((op.loss(xr, y) * w).sum(), xr)[0]
    assert_close(ga, g * w.view(-1, 1, 1, 1, 1), 1e-5, "autograd through loss")
    gv = torch.autograd.grad((op(xr) * v).sum(), xr)[0]
    assert_close(gv, op._type1(v, 1.0), 1e-6, "autograd through A(x)")
    # deterministic (the spread is a gather) and chunking-invariant
    assert torch.equal(op.gradient(x, y), g)
    op.max_chunk = 2
    assert torch.equal(op.gradient(x, y), g)


def test_radial_langevin_step_runs_in_the_sampler(dev):
    from step_amd.sampler import MCMCSampler
    op = make(spokes=16, frames=12)
    gen = torch.Generator(device=dev).manual_seed(2)
    x0 = torch.randn(2, 12, 2, 192, 192, device=dev, generator=gen)
    y = op(torch.randn(1, 12, 2, 192, 192, device=dev, generator=gen))
    s = MCMCSampler(num_steps=3, lr=1e-7, tau=1.0, mc_algo="langevin", prior_solver="gaussian")
    l0 = op.loss(x0, y)
    out = s.optimize(x0, op, y, 10.0, 0.0)
    assert torch.isfinite(out).all() and (op.loss(out, y) < l0).all()


def test_fused_grid_transforms_equal_pad_fft_crop(dev):
    """nufft_fwd_grid == nufft_pad_apod + fft2 and nufft_inv_grid == unnormalised ifft2 +
    nufft_crop_apod on the whole grid (not only where samples look), to fp32 FFT rounding."""
    op = make(spokes=8, frames=12)
    assert op._own_fft()
    gen = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(3, 12, 2, 192, 192, device=dev, generator=gen)
    op.grid_fft = "fused"
    a = op._grid_fft(x)
    op.grid_fft = "library"
    b = op._grid_fft(x)
    assert_close(torch.view_as_real(a), torch.view_as_real(b), 2e-6, "forward grid")
    # adjoint half on an arbitrary grid: reach it through _type1's second half
    from step_amd import _lib
    import ctypes as C
    grid = torch.randn(3, 12, 384, 384, 2, device=dev, generator=gen)
    cs = torch.tensor([0.5, -1.0, 2.0], device=dev)
    plan, st = op._plan(x.device), _lib.stream_of(x)
    out_f = torch.empty_like(x)
    work = torch.empty(3, 12, 192, 384, 2, device=dev)
    _lib.check(_lib.lib().nufft_inv_grid(grid.data_ptr(), 36, 12, C.byref(plan), cs.data_ptr(), 0.25, work.data_ptr(),
                                         out_f.data_ptr(), st), "nufft_inv_grid")
    Z = torch.fft.ifft2(torch.view_as_complex(grid), norm="forward")
    out_l = torch.empty_like(x)
    _lib.check(_lib.lib().nufft_crop_apod(Z.data_ptr(), 36, 12, C.byref(plan), cs.data_ptr(), 0.25, out_l.data_ptr(), st),
               "nufft_crop_apod")
    assert_close(out_f, out_l, 2e-6, "adjoint image")
    rc = _lib.lib().nufft_fwd_grid(x.data_ptr(), 36, 5, C.byref(plan), work.data_ptr(), a.data_ptr(), st)
    assert rc != 0 and b"frames" in _lib.lib().step_last_error()


@pytest.mark.parametrize("shared", [False, True])
def test_tiled_interpolation_equals_the_gathering_kernel(shared, dev):
    """nufft_interp out of LDS tiles (owner bins in the plan) against the same plan without the
    bins (every sample gathers from global memory): same products in the same order."""
    from step_amd.forward_operator import golden_angle_radial
    traj = golden_angle_radial(1 if shared else 12, 24, 192)
    rng = np.random.RandomState(3)
    extra = rng.uniform(-96, 96, (traj.shape[0], 700, 2))
    extra[:, :4] = [[-96, -96], [96, 96], [-96, 95.99], [0, 0]]          # corners: footprints that wrap
    traj = np.concatenate([traj, extra], axis=1)
    a, b = make(spokes=24, frames=12, trajectory=traj), make(spokes=24, frames=12, trajectory=traj)
    for k in ("own_ptr", "own_ent", "own_x0", "own_y0", "own_wx", "own_wy"):
        b._host[k] = None
    gen = torch.Generator(device=dev).manual_seed(9)
    x = torch.randn(5, 12, 2, 192, 192, device=dev, generator=gen)           # 5 chains: a ragged group of 4 + 1
    ya = a(x)
    assert_close(ya, b(x), 1e-6, "samples")
    y = ya[:1] + 0.1 * torch.randn(1, 12, a.J, 2, device=dev, generator=gen)
    (ga, la), (gb, lb) = a.gradient(x, y, return_loss=True), b.gradient(x, y, return_loss=True)
    assert_close(la, lb, 1e-6, "loss")
    assert_close(ga, gb, 1e-6, "gradient")
    assert_close(a.loss(x, y), b.loss(x, y), 1e-6, "per-chain loss")
    yb = ya + 0.1 * torch.randn(ya.shape, device=dev, generator=gen)             # a measurement per chain
    assert_close(a.loss(x, yb), b.loss(x, yb), 1e-6, "per-chain loss, per-chain measurement")


@pytest.mark.parametrize("route", ["generic", "library"])
@pytest.mark.parametrize("R", [64, 80, 104, 256])
def test_other_resolutions(R, route, dev):
    """R != 192: pad / crop kernels around the library's own any-size FFT2 (nufft_fft2_any, mixed-radix passes
    over LDS: the default) or around torch.fft (forced); G = 128 / 160 / 512 have whole 32 x 32 tiles
    (interpolation out of LDS), G = 208 = 16 x 13 has not and takes a direct 13-point pass -- all against the
    direct NUDFT."""
    from step_amd.forward_operator import golden_angle_radial
    frames = 3
    traj = golden_angle_radial(frames, 9, R)
    op, orc = make(spokes=9, frames=frames, resolution=R, trajectory=traj), NUDFTOracle(traj, R)
    assert not op._own_fft() and op._any_fft()
    if route == "library":
        op.grid_fft = "library"
        assert not op._any_fft()
    gen = torch.Generator().manual_seed(R)
    x = torch.randn(2, frames, 2, R, R, generator=gen)
    y = orc(torch.randn(1, frames, 2, R, R, generator=gen)).float()
    assert_close(op(x.to(dev)), orc(x), RTOL, "type 2")
    g_ref, l_ref = orc.gradient(x, y, return_loss=True)
    g_got, l_got = op.gradient(x.to(dev), y.to(dev), return_loss=True)
    assert_close(l_got, l_ref, RTOL, "loss")
    assert_close(g_got, g_ref, RTOL, "gradient")
    # 16 chains take the lane-per-chain spread (any G with whole 16 x 2 tiles): the bits of the tile kernel
    v = torch.randn(16, frames, op.J, 2, generator=gen).to(dev)
    wide = op._type1(v, 1.0)
    assert torch.equal(wide[3:4], op._type1(v[3:4], 1.0)) and torch.equal(wide[15:], op._type1(v[15:], 1.0))
    assert not op._fused_fwd(dev)
    op.grid_fft = "fused"
    with pytest.raises(RuntimeError, match="built for R = 192"):
        op.gradient(x.to(dev), y.to(dev))


@pytest.mark.parametrize("G", [2, 24, 128, 160, 194, 200, 384, 486, 512, 1024])
def test_any_size_grid_fft2_against_the_library(G, dev):
    """nufft_fft2_any through the C ABI: in place, both signs, against torch.fft on the same grids."""
    from step_amd import _lib
    L = _lib.lib()
    assert L.nufft_fft2_any_supported(G) and not L.nufft_fft2_any_supported(1025) and not L.nufft_fft2_any_supported(1)
    gen = torch.Generator(device=dev).manual_seed(G)
    z = torch.randn(3, G, G, 2, device=dev, generator=gen)
    zc = torch.view_as_complex(z.clone())
    for sign, want in ((-1, torch.fft.fft2(zc)), (1, torch.fft.ifft2(zc, norm="forward"))):
        w = z.clone()
        _lib.check(L.nufft_fft2_any(w.data_ptr(), 3, G, sign, _lib.stream_of(w)), "nufft_fft2_any")
        assert_close(w, torch.view_as_real(want), 2e-6, "sign %d" % sign)
    assert L.nufft_fft2_any(z.data_ptr(), 3, G, 0, _lib.stream_of(z)) != 0
    assert L.nufft_fft2_any(z.data_ptr(), 3, 2048, -1, _lib.stream_of(z)) != 0


@pytest.mark.parametrize("shared", [False, True])
def test_lane_per_chain_spread_equals_the_tile_kernel(shared, dev):
    """nufft_spread from 16 chains on (a lane per chain, 16 x 2 tiles, weights from the plan's table)
    against the same plan without those lists (8 chains per 16 x 4 tile): the same terms in the same
    order, bit for bit; ragged chain counts, footprints that wrap, more than one group of 64; and
    against the direct NUDFT's adjoint."""
    import ctypes as C
    from step_amd import _lib
    from step_amd.forward_operator import golden_angle_radial
    frames = 3 if shared else 12
    traj = golden_angle_radial(1 if shared else frames, 24, 192)
    rng = np.random.RandomState(4)
    extra = rng.uniform(-96, 96, (traj.shape[0], 300, 2))
    extra[:, :4] = [[-96, -96], [96, 96], [-96, 95.99], [0, 0]]
    traj = np.concatenate([traj, extra], axis=1)
    a, b = make(spokes=24, frames=frames, trajectory=traj), make(spokes=24, frames=frames, trajectory=traj)
    assert a._host["wide_rec"] is not None and (a._host["wide_rec"][:, :, 2] % 4 == 0).all()
    for k in ("wide_rec", "wide_ent", "wide_w"):
        b._host[k] = None
    gen = torch.Generator(device=dev).manual_seed(11)
    L = _lib.lib()
    for B in (16, 37, 70):
        v = torch.randn(B, frames, a.J, 2, device=dev, generator=gen)
        ga = torch.full((B, frames, 384, 384, 2), float("nan"), device=dev)
        gb = torch.full((B, frames, 384, 384, 2), float("nan"), device=dev)
        for op, g in ((a, ga), (b, gb)):
            _lib.check(L.nufft_spread(v.data_ptr(), B * frames, frames, C.byref(op._plan(v.device)), g.data_ptr(),
                                      _lib.stream_of(v)), "nufft_spread")
        assert torch.isfinite(ga).all()
        assert torch.equal(ga, gb), "B = %d: %g" % (B, (ga - gb).abs().max().item())
    if shared:
        v = torch.randn(17, frames, a.J, 2, device=dev, generator=gen)
        got = a._type1(v, 1.0)
        assert torch.equal(got, b._type1(v, 1.0))
        orc = NUDFTOracle(traj, 192)
        assert_close(got[[0, 16]], orc.adjoint(v[[0, 16]].cpu()), RTOL, "type 1 at 17 chains")


@pytest.mark.parametrize("shared", [False, True])
def test_fused_column_pass_and_interpolation_equals_the_grid_route(shared, dev):
    """nufft_fwd_interp (the forward column pass interpolates from its own LDS strip, no grid in HBM)
    against nufft_fwd_grid + nufft_interp: the same sums over the same transformed values, to fp32
    rounding; footprints that wrap in x (the last strip reaches into strip 0) and in y."""
    from step_amd.forward_operator import golden_angle_radial
    traj = golden_angle_radial(1 if shared else 12, 24, 192)
    rng = np.random.RandomState(6)
    extra = rng.uniform(-96, 96, (traj.shape[0], 700, 2))
    extra[:, :5] = [[-96, -96], [96, 96], [-96, 95.99], [0, 0], [95.2, -0.3]]
    traj = np.concatenate([traj, extra], axis=1)
    a, b = make(spokes=24, frames=12, trajectory=traj), make(spokes=24, frames=12, trajectory=traj)
    a.fwd_interp, b.fwd_interp = "fused", "grid"
    assert a._fused_fwd(dev) == 24 and b._fused_fwd(dev) == 0
    gen = torch.Generator(device=dev).manual_seed(13)
    x = torch.randn(3, 12, 2, 192, 192, device=dev, generator=gen)
    ya = a(x)
    assert_close(ya, b(x), 2e-6, "samples")
    y = ya[:1] + 0.1 * torch.randn(1, 12, a.J, 2, device=dev, generator=gen)
    (ga, la), (gb, lb) = a.gradient(x, y, return_loss=True), b.gradient(x, y, return_loss=True)
    assert_close(la, lb, 2e-6, "loss")
    assert_close(ga, gb, 2e-6, "gradient")
    yb = ya + 0.1 * torch.randn(ya.shape, device=dev, generator=gen)             # a measurement per chain
    assert_close(a.loss(x, yb), b.loss(x, yb), 2e-6, "per-chain loss, per-chain measurement")
    assert torch.equal(a.gradient(x, y), ga)


@pytest.mark.parametrize("shared", [False, True])
def test_adjoint_that_skips_the_empty_tiles_equals_spread_plus_inverse(shared, dev):
    """nufft_adjoint (the spread leaves the tiles outside the per-column row ranges unwritten, the inverse
    column pass takes them as zero) against nufft_spread + nufft_inv_grid: bit for bit, with the scratch
    grid pre-filled with NaN (nothing unwritten may be read)."""
    import ctypes as C
    from step_amd import _lib
    from step_amd.forward_operator import golden_angle_radial
    frames = 12
    traj = golden_angle_radial(1 if shared else frames, 24, 192)
    if shared:                                              # a trajectory that leaves whole tile columns empty
        traj = traj[:, (np.abs(traj[0, :, 0]) < 60) & (np.abs(traj[0, :, 1]) < 90)]
    op = make(spokes=24, frames=frames, trajectory=traj)
    rec, rng = op._host["wide_rec"], op._host["wide_range"]
    assert rec[:, :, 3].mean() > 0.15 and ((rec[:, :, 3] == 1) <= (rec[:, :, 2] == 0)).all()
    if shared:
        assert (rng[0, :, 0] > rng[0, :, 1]).any()          # empty columns: lo > hi
    L, plan = _lib.lib(), op._plan(torch.device(dev))
    gen = torch.Generator(device=dev).manual_seed(21)
    for B in (16, 37):
        assert L.nufft_adjoint_supported(C.byref(plan), B)
        v = torch.randn(B, frames, op.J, 2, device=dev, generator=gen)
        cs = torch.rand(B, device=dev, generator=gen) + 0.5
        st = _lib.stream_of(v)
        out_a, out_b = torch.empty(B, frames, 2, 192, 192, device=dev), torch.empty(B, frames, 2, 192, 192, device=dev)
        work = torch.empty(B, frames, 192, 384, 2, device=dev)
        grid = torch.full((B, frames, 384, 384, 2), float("nan"), device=dev)
        _lib.check(L.nufft_adjoint(v.data_ptr(), B * frames, frames, C.byref(plan), rng.ctypes.data, cs.data_ptr(), 0.5,
                                   grid.data_ptr(), work.data_ptr(), out_a.data_ptr(), st), "nufft_adjoint")
        assert torch.isnan(grid).any()                      # something was skipped
        _lib.check(L.nufft_spread(v.data_ptr(), B * frames, frames, C.byref(plan), grid.data_ptr(), st), "nufft_spread")
        _lib.check(L.nufft_inv_grid(grid.data_ptr(), B * frames, frames, C.byref(plan), cs.data_ptr(), 0.5, work.data_ptr(),
                                    out_b.data_ptr(), st), "nufft_inv_grid")
        assert torch.isfinite(out_a).all() and torch.equal(out_a, out_b)
    assert not L.nufft_adjoint_supported(C.byref(plan), 8)
    a, b = op._type1(v, 1.0), None
    op.adjoint = "full"
    b = op._type1(v, 1.0)
    assert torch.equal(a, b)


@pytest.mark.parametrize("width", [2, 4, 8])
def test_other_kernel_widths_agree_between_the_routes(width, dev):
    """Kernel widths other than the default 6 (the fused forward stages 16 + W - 1 columns, the spread tables
    hold W x W footprints): fused forward == grid route, lane-per-chain spread == tile spread bit for bit,
    skipping adjoint == full adjoint bit for bit; width 8 also against the direct NUDFT."""
    from step_amd.forward_operator import golden_angle_radial
    traj = golden_angle_radial(12, 16, 192)
    a = make(spokes=16, frames=12, trajectory=traj, kernel_width=width)
    b = make(spokes=16, frames=12, trajectory=traj, kernel_width=width)
    b.fwd_interp, b.adjoint = "grid", "full"
    for k in ("wide_rec", "wide_ent", "wide_w"):
        b._host[k] = None
    gen = torch.Generator(device=dev).manual_seed(width)
    x = torch.randn(17, 12, 2, 192, 192, device=dev, generator=gen)
    ya = a(x)
    assert_close(ya, b(x), 2e-6, "samples")
    v = torch.randn(17, 12, a.J, 2, device=dev, generator=gen)
    assert torch.equal(a._type1(v, 1.0), b._type1(v, 1.0))
    y = ya[:1] + 0.1 * torch.randn(1, 12, a.J, 2, device=dev, generator=gen)
    (ga, la), (gb, lb) = a.gradient(x, y, return_loss=True), b.gradient(x, y, return_loss=True)
    assert_close(la, lb, 2e-6, "loss")
    assert_close(ga, gb, 2e-6, "gradient")
    if width == 8:
        orc = NUDFTOracle(traj, 192)
        assert_close(ya[[0, 16]], orc(x[[0, 16]].cpu()), RTOL, "type 2 at width 8")
