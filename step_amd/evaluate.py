"""Evaluators behind the reference's interface (reference evaluate.py:8-53): a name registry
(``register_evaluator`` / ``get_evaluator(name, op=...)``), objects called as
``evaluator(x, y, xhat) -> {metric: tensor}`` with a ``main_eval_fn_name``.  ``LatentDAPS.sample``
calls them twice per annealing step (daps.py:326-327), ``step_amd.run_task`` once per sample
(sample.py:71).

'blackhole' (evaluate.py:43-53): the larger of the normalised closure-phase and
log-closure-amplitude chi^2 of ``xhat`` -- all of it on the operator's kernels
(``evaluate_chisq``: DFT with the clip fused into the load, ``bh_closure_chisq``).

'mri' (evaluate.py:26-40): PSNR and SSIM between the min-max normalised first videos of ``x`` and
``xhat``, frames as the batch.  The reference takes both from ``piq`` (requirements.txt; not
installed here, not vendored): ``psnr`` / ``ssim`` below restate piq's documented definitions
(PSNR = -10 log10(MSE + 1e-8) per frame; SSIM with an 11 x 11 Gaussian window of sigma 1.5,
k1 = 0.01, k2 = 0.03, valid convolution, average-pool down-sampling by round(min(H, W) / 256)
when that exceeds 1; both averaged over frames) -- PARITY UNPINNED: nothing in the reference
holds a value to check them against.  Plain torch: metrics are not hot-path work."""
import torch
import torch.nn.functional as F

from ._registry import Registry

__all__ = ["register_evaluator", "get_evaluator", "MRIEvaluator", "BlackholeEvaluator", "psnr", "ssim"]

__EVALUATOR__ = Registry("evaluator")
register_evaluator = __EVALUATOR__.register
get_evaluator = __EVALUATOR__.create


def _check_pair(x, y, data_range):
    if x.shape != y.shape or x.dim() != 4:
        raise ValueError("expected two [N, C, H, W] tensors of one shape, got %s and %s" % (tuple(x.shape), tuple(y.shape)))
    lo, hi = min(x.min().item(), y.min().item()), max(x.max().item(), y.max().item())
    if lo < 0 or hi > data_range:
        raise ValueError("values outside [0, %g]: [%g, %g]" % (data_range, lo, hi))


def psnr(x, y, data_range=1.0):
    """Peak signal-to-noise ratio per image, averaged over the batch."""
    _check_pair(x, y, data_range)
    mse = ((x / data_range - y / data_range) ** 2).flatten(1).mean(dim=1)
    return (-10.0 * torch.log10(mse + 1e-8)).mean()


def _gauss_window(size, sigma, channels, like):
    c = torch.arange(size, dtype=like.dtype, device=like.device) - (size - 1) / 2.0
    g = torch.exp(-(c ** 2) / (2.0 * sigma ** 2))
    w = torch.outer(g, g)
    return (w / w.sum()).expand(channels, 1, size, size).contiguous()


def ssim(x, y, data_range=1.0, kernel_size=11, kernel_sigma=1.5, k1=0.01, k2=0.03, downsample=True):
    """Structural similarity (Wang et al. 2004) per channel on Gaussian-windowed statistics,
    averaged over window positions, channels and the batch."""
    _check_pair(x, y, data_range)
    x, y = x / data_range, y / data_range
    f = max(1, round(min(x.shape[-2:]) / 256))
    if downsample and f > 1:
        x, y = F.avg_pool2d(x, f), F.avg_pool2d(y, f)
    ch = x.shape[1]
    w = _gauss_window(kernel_size, kernel_sigma, ch, x)
    mean = lambda t: F.conv2d(t, w, groups=ch)
    mx, my = mean(x), mean(y)
    vx, vy, cxy = mean(x * x) - mx * mx, mean(y * y) - my * my, mean(x * y) - mx * my
    c1, c2 = k1 ** 2, k2 ** 2
    contrast = (2.0 * cxy + c2) / (vx + vy + c2)
    value = (2.0 * mx * my + c1) / (mx * mx + my * my + c1) * contrast
    return value.mean(dim=(-1, -2)).mean(dim=1).mean()


@register_evaluator('mri')
class MRIEvaluator:
    def __init__(self, op):
        self.main_eval_fn_name = 'psnr'
        self.op = op

    def norm(self, x):
        return (x - x.min()) / (x.max() - x.min())

    def __call__(self, x, y, xhat):
        a, b = self.norm(x[0]), self.norm(xhat[0])            # [T, C, H, W]: frames as the batch
        return {'psnr': psnr(a, b, data_range=1.0), 'ssim': ssim(a, b, data_range=1.0)}


@register_evaluator('blackhole')
class BlackholeEvaluator:
    def __init__(self, op):
        self.main_eval_fn_name = 'chisq'
        self.op = op

    def __call__(self, x, y, xhat):
        chi = self.op.evaluate_chisq(xhat, y, chi_sq_list=['cphase', 'logcamp'], normalize=True)
        return {self.main_eval_fn_name: torch.max(chi['cphase'], chi['logcamp'])}
