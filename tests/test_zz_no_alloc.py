"""mri_loss_grad_static -- the entry point that allocates nothing (include/step_amd.h) -- against
mri_loss_grad through the C ABI: bit-identical loss, total and gradient on two calls in a row,
the second with a smaller input (the static workspace keeps the first call's contents)."""
import ctypes as C

import pytest
import torch

from step_amd import _lib
from step_amd.forward_operator import get_operator

pytestmark = pytest.mark.gpu


def _both(op, x, y):
    L = _lib.lib()
    B, T, R = x.shape[0], x.shape[1], x.shape[3]
    rows, nrows = op._tables(x.device)
    st = _lib.stream_of(x)
    outs = []
    for static in (False, True):
        loss = torch.full((B + 1,), float("nan"), device=x.device)
        grad = torch.full_like(x, float("nan"))
        if static:
            before = torch.cuda.memory_allocated(x.device)
            rc = L.mri_loss_grad_static(x.data_ptr(), y.data_ptr(), y.shape[0], B, T, R, rows.data_ptr(),
                                        nrows.data_ptr(), op.mask.shape[1], op._max_rows, None,
                                        loss.data_ptr(), loss.data_ptr() + 4 * B, grad.data_ptr(), st)
            assert torch.cuda.memory_allocated(x.device) == before
        else:
            part = torch.empty(2 * B * T, device=x.device)
            rc = L.mri_loss_grad(x.data_ptr(), y.data_ptr(), y.shape[0], B, T, R, rows.data_ptr(),
                                 nrows.data_ptr(), op.mask.shape[1], op._max_rows, None, 0,
                                 part.data_ptr(), loss.data_ptr(), loss.data_ptr() + 4 * B,
                                 grad.data_ptr(), st)
        _lib.check(rc, "mri_loss_grad%s" % ("_static" if static else ""))
        torch.cuda.synchronize()
        outs.append((loss.clone(), grad.clone()))
    return outs


@pytest.mark.parametrize("mode,y_per_chain", [("same_mask", False), ("split_mask", False), ("same_mask", True)])
def test_static_workspace_matches_caller_workspace(mode, y_per_chain):
    dev = torch.device("cuda:0")
    op = get_operator("mri", acc_factor=6, mode=mode, resolution=192, frames=12, num_acs_lines=24)
    g = torch.Generator(device=dev).manual_seed(5)
    for B in (20, 3):                                    # second call smaller than the first
        x = torch.randn(B, 12, 2, 192, 192, device=dev, generator=g)
        y = op(torch.randn(B if y_per_chain else 1, 12, 2, 192, 192, device=dev, generator=g))
        (l0, g0), (l1, g1) = _both(op, x, y)
        assert torch.isfinite(l0).all() and torch.isfinite(g0).all()
        assert torch.equal(l0, l1) and torch.equal(g0, g1)


def test_static_workspace_limit_is_an_error():
    dev = torch.device("cuda:0")
    op = get_operator("mri", acc_factor=6, mode="same_mask", resolution=192, frames=12, num_acs_lines=24)
    rows, nrows = op._tables(dev)
    one = torch.zeros(8, device=dev)
    # 2731 * 12 = 32772 frames > 32768: rejected before any pointer is touched
    rc = _lib.lib().mri_loss_grad_static(one.data_ptr(), one.data_ptr(), 1, 2731, 12, 192, rows.data_ptr(),
                                         nrows.data_ptr(), 1, op._max_rows, None, one.data_ptr(), None,
                                         None, _lib.stream_of(one))
    assert rc == -1
    assert b"exceeds the static workspace" in _lib.lib().step_last_error()
