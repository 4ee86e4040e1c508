"""CPU: the evaluator registry (reference evaluate.py:8-24) and the MRI metrics.  PSNR / SSIM
restate piq's definitions (piq is absent: PARITY UNPINNED); they are checked here against a direct
evaluation of the formulas in float64 and through their defining properties."""
import numpy as np
import pytest
import torch

from step_amd.evaluate import MRIEvaluator, get_evaluator, psnr, register_evaluator, ssim


def test_registry():
    ev = get_evaluator("mri", op="the operator")
    assert isinstance(ev, MRIEvaluator) and ev.op == "the operator" and ev.name == "mri"
    assert ev.main_eval_fn_name == "psnr"
    assert get_evaluator("blackhole", op=None).main_eval_fn_name == "chisq"
    with pytest.raises(NameError):
        get_evaluator("nope", op=None)
    with pytest.warns(UserWarning):
        @register_evaluator("mri")
        class Other:                                          # a second class under a taken name
            def __init__(self, op): pass
    with pytest.warns(UserWarning):
        register_evaluator("mri")(MRIEvaluator)               # put the real one back


def ssim_direct(x, y, size=11, sigma=1.5, k1=0.01, k2=0.03):
    """Window by window, in float64 (x, y: [H, W])."""
    c = np.arange(size) - (size - 1) / 2
    g = np.exp(-c ** 2 / (2 * sigma ** 2))
    w = np.outer(g, g)
    w /= w.sum()
    H, W = x.shape
    vals = []
    for i in range(H - size + 1):
        for j in range(W - size + 1):
            a, b = x[i:i + size, j:j + size], y[i:i + size, j:j + size]
            ma, mb = (w * a).sum(), (w * b).sum()
            va, vb, cab = (w * a * a).sum() - ma * ma, (w * b * b).sum() - mb * mb, (w * a * b).sum() - ma * mb
            vals.append((2 * ma * mb + k1 ** 2) / (ma * ma + mb * mb + k1 ** 2) * (2 * cab + k2 ** 2) / (va + vb + k2 ** 2))
    return np.mean(vals)


def test_psnr_and_ssim_against_the_formulas():
    rng = np.random.RandomState(0)
    x = rng.rand(3, 2, 20, 17)
    y = np.clip(x + 0.1 * rng.randn(*x.shape), 0, 1)
    tx, ty = torch.from_numpy(x), torch.from_numpy(y)
    mse = ((x - y) ** 2).reshape(3, -1).mean(1)
    assert abs(psnr(tx, ty).item() - (-10 * np.log10(mse + 1e-8)).mean()) < 1e-9
    want = np.mean([[ssim_direct(x[n, c], y[n, c]) for c in range(2)] for n in range(3)])
    assert abs(ssim(tx, ty).item() - want) < 1e-9
    # float32 inputs, as the sampler hands them over
    assert abs(ssim(tx.float(), ty.float()).item() - want) < 1e-5


def test_metric_properties():
    g = torch.Generator().manual_seed(1)
    x = torch.rand(4, 2, 48, 48, generator=g)
    assert abs(ssim(x, x).item() - 1) < 1e-6 and psnr(x, x).item() == pytest.approx(80.0, abs=1e-4)
    noisy = [(x + s * torch.randn(x.shape, generator=g)).clamp(0, 1) for s in (0.02, 0.1, 0.3)]
    s_vals, p_vals = [ssim(x, n).item() for n in noisy], [psnr(x, n).item() for n in noisy]
    assert s_vals[0] > s_vals[1] > s_vals[2] and p_vals[0] > p_vals[1] > p_vals[2]
    assert ssim(x, noisy[1]).item() == pytest.approx(ssim(noisy[1], x).item(), abs=1e-6)
    with pytest.raises(ValueError):
        psnr(x, x + 1.0)
    with pytest.raises(ValueError):
        ssim(x, x[:, :1])
    # large frames are average-pooled first (factor round(min(H, W) / 256))
    big = torch.rand(1, 1, 512, 600, generator=g)
    pooled = torch.nn.functional.avg_pool2d(big, 2)
    assert ssim(big, big.flip(-1)).item() == pytest.approx(ssim(pooled, pooled.flip(-1), downsample=False).item(), abs=1e-6)


def test_mri_evaluator_takes_the_first_video_and_normalises():
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 12, 2, 32, 32, generator=g)
    xhat = x + 0.05 * torch.randn(x.shape, generator=g)
    ev = MRIEvaluator(op=None)
    out = ev(x, None, xhat)
    assert set(out) == {"psnr", "ssim"} and 0 < out["ssim"].item() < 1 and out["psnr"].item() > 20
    # only video 0 enters, and an affine change of xhat's range is normalised away
    xhat2 = xhat.clone()
    xhat2[1] = 0
    assert ev(x, None, xhat2)["psnr"].item() == pytest.approx(out["psnr"].item(), abs=1e-4)
    assert ev(x, None, torch.stack([3 * xhat[0] + 1, xhat[1]]))["psnr"].item() == pytest.approx(out["psnr"].item(), abs=1e-3)
