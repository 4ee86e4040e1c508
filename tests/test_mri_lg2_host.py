"""CPU: the body of the pipelined MRI loss+gradient kernel (step_amd/csrc/mri_lg2.h) is plain C++
over a small context, so the very code the GPU runs is executed here on the host -- one pthread
per GPU thread, __syncthreads = pthread_barrier, LDS = a NaN-filled heap block
(tests/host_emu/mri_lg2_emu.cpp) -- and compared with the oracle: index arithmetic, barrier
placement (a missing barrier reads NaN), sign folding, both variants of the masked-out rows' sum."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle.mri_oracle import MRIOracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    out = tmp_path_factory.mktemp("mri_lg2_emu") / "libemu.so"
    subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-pthread",
                    "-I", os.path.join(ROOT, "step_amd", "csrc"),
                    os.path.join(ROOT, "tests", "host_emu", "mri_lg2_emu.cpp"), "-o", str(out)], check=True)
    return C.CDLL(str(out))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def run_case(emu, mode, acc, acs, B, T, yb, with_grad=True, scale=None, seed=0):
    orc = MRIOracle(acc, mode, 192, T, acs)
    m = orc.mask[0, :, 0, :, 0].numpy()
    mf = m.shape[0]
    rows = np.full((mf, 192), -1, np.int32)
    nrows = np.zeros(mf, np.int32)
    for f in range(mf):
        k = np.nonzero(m[f])[0].astype(np.int32)
        rows[f, :len(k)] = k
        nrows[f] = len(k)
    assert nrows.max() <= 64
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, 2, 192, 192, generator=g)
    y = orc(torch.randn(yb, T, 2, 192, 192, generator=g)) + 0.1 * torch.randn(yb, T, 2, 192, 192, generator=g)
    g_ref = orc.gradient(x, y)
    l_ref = orc.loss(x, y).numpy()
    xn, yn = x.numpy().copy(), y.numpy().copy()
    part = np.full(B * T, np.nan, np.float32)
    unk = np.full(B * T, np.nan, np.float32)
    grad = np.full_like(xn, np.nan) if with_grad else None
    cs = None if scale is None else np.asarray(scale, np.float32)
    rc = emu.lg2_emu(_ptr(xn), _ptr(yn), yb, _ptr(rows), _ptr(nrows), mf, T, B * T, int(nrows.max()),
                     _ptr(cs), _ptr(part), _ptr(unk), _ptr(grad), B * T)
    assert rc == 0
    loss = part.reshape(B, T).sum(1)
    if yb == 1:                                  # shares of the masked-out rows, the first 16 chains'
        loss = loss + unk[:min(B, 16) * T].sum()
        assert np.isnan(unk[min(B, 16) * T:]).all()
    assert np.abs(loss - l_ref).max() <= 1e-5 * np.abs(l_ref).max()
    if with_grad:
        ref = g_ref.numpy()
        if scale is not None:
            ref = ref * cs.reshape(-1, 1, 1, 1, 1)
        assert np.abs(grad - ref).max() <= 1e-5 * np.abs(ref).max()


def test_same_mask_shared_measurement(emu):
    run_case(emu, "same_mask", 6, 24, B=1, T=3, yb=1)


def test_split_mask_measurement_per_chain_with_chain_scale(emu):
    run_case(emu, "split_mask", 6, 24, B=2, T=2, yb=2, scale=[0.5, -2.0])


def test_random_split_mask(emu):
    run_case(emu, "random_split_mask", 8, 12, B=1, T=3, yb=1)


def test_loss_only(emu):
    run_case(emu, "same_mask", 6, 24, B=1, T=2, yb=1, with_grad=False)


def test_full_row_buffer(emu):
    run_case(emu, "same_mask", 3, 0, B=1, T=2, yb=1)          # 64 kept rows


def test_more_chains_than_shares(emu):
    run_case(emu, "same_mask", 6, 24, B=18, T=1, yb=1, seed=3)  # 16 shares, chains 16 and 17 hold none


def test_frames_without_kept_rows(emu):
    run_case(emu, "split_mask", 192, 0, B=2, T=3, yb=1)       # one kept row in all: two empty mask frames
