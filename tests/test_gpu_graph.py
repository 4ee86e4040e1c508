"""GPU: the operator steps can be captured into a HIP graph and replayed (the C ABI allocates
nothing and never synchronises; the host layer's tensors come from torch's capture-aware
allocator), and a replay on new input values equals the eager call."""
import pytest
import torch

from tests.helpers import RTOL, assert_close

pytestmark = pytest.mark.gpu


def _capture_and_compare(op, x, y, dev):
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):                       # warm-up on the side stream, as torch asks
        for _ in range(3):
            op.gradient(x, y, return_loss=True)
    torch.cuda.current_stream(dev).wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        g_static, l_static = op.gradient(x, y, return_loss=True)
    x_new = torch.randn_like(x) * 0.5
    g_eager, l_eager = op.gradient(x_new, y, return_loss=True)
    x.copy_(x_new)
    graph.replay()
    torch.cuda.synchronize(dev)
    assert_close(g_static, g_eager, RTOL, "graph replay gradient")
    assert_close(l_static, l_eager, RTOL, "graph replay loss")


def test_mri_step_in_a_graph(dev):
    from step_amd.forward_operator import get_operator
    op = get_operator("mri", acc_factor=6, mode="same_mask", resolution=192, frames=12, num_acs_lines=24)
    gen = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(4, 12, 2, 192, 192, device=dev, generator=gen)
    y = op(torch.randn(1, 12, 2, 192, 192, device=dev, generator=gen))
    _capture_and_compare(op, x, y, dev)


def test_blackhole_step_in_a_graph(dev):
    from step_amd import eht_lite
    from step_amd.forward_operator import BlackHoleImagingTorch
    tab = eht_lite.synthetic_table(6, 64, seed=5, sigma_scale=5e-3)
    op = BlackHoleImagingTorch(imsize=64, w_vis=0, w_amp=0, w_cphase=1, w_logcamp=1, w_flux=0.26,
                               loss_normalize=True, ref_multiplier=0.001, device=dev, obs_table=tab)
    gen = torch.Generator(device=dev).manual_seed(6)
    x = torch.rand(3, 64, 1, 64, 64, device=dev, generator=gen) * 2 - 1
    y = op(torch.rand(1, 64, 1, 64, 64, device=dev, generator=gen) * 2 - 1)
    _capture_and_compare(op, x, y, dev)
