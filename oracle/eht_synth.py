"""Oracle (TEST INFRASTRUCTURE, not product).  PARITY UNPINNED.

Restates, from memory, the few pieces of ``ehtim==1.2.10`` (requirements.txt:2;
not vendored under /root/reference, not installed here) that the reference calls
at set-up time (blackhole.py:188-256): ``obsh.ftmatrix`` with
``trianglePulse2D``, and the minimal triangle / quadrangle sets behind
``Obsdata.c_phases()`` / ``c_amplitudes()``.  Used to build synthetic operators
of the reference's shapes and to cross-check the product's
separable factor construction against a dense evaluation of the same formula.
"""
import numpy as np

PSIZE = 2.4240684055470684e-12      # blackhole.py:190


def tri_pulse_ft(omega, pdim):
    """1-D factor of trianglePulse2D(..., dom='F'): 1 at 0 else 4/(pdim^2 w^2) sin^2(pdim w/2)."""
    omega = np.asarray(omega, dtype=np.float64)
    out = np.ones_like(omega)
    nz = omega != 0
    out[nz] = 4.0 / (pdim ** 2 * omega[nz] ** 2) * np.sin(pdim * omega[nz] / 2.0) ** 2
    return out


def ftmatrix_dense(pdim, xdim, ydim, uv):
    """Dense DFT matrix [F, ydim*xdim] complex128, evaluated pixel by pixel:
    A[f, j*xdim+i] = pulse(2 pi u, 2 pi v) exp(2 pi i (y_j v + x_i u)),
    x_i = (-i + xdim/2 - 1/2) pdim (same for y)."""
    uv = np.asarray(uv, dtype=np.float64)
    xs = (np.arange(0, -xdim, -1) * pdim + (pdim * xdim) / 2.0 - pdim / 2.0)
    ys = (np.arange(0, -ydim, -1) * pdim + (pdim * ydim) / 2.0 - pdim / 2.0)
    yy, xx = np.meshgrid(ys, xs, indexing="ij")            # [ydim, xdim]
    u = uv[:, 0][:, None, None]
    v = uv[:, 1][:, None, None]
    pulse = tri_pulse_ft(2 * np.pi * uv[:, 0], pdim) * tri_pulse_ft(2 * np.pi * uv[:, 1], pdim)
    ph = 2j * np.pi * (yy[None] * v + xx[None] * u)
    return (pulse[:, None, None] * np.exp(ph)).reshape(uv.shape[0], ydim * xdim)


def minimal_triangles(sites):
    """(N-1)(N-2)/2 triangles through the reference site sites[0]."""
    ref, rest = sites[0], list(sites[1:])
    return [(ref, rest[i], rest[j]) for i in range(len(rest)) for j in range(i + 1, len(rest))]


def minimal_quadrangles(sites):
    """N(N-3)/2 quadrangles, convention (12)(34)/(14)(23)."""
    ref, s = sites[0], list(sites)
    quads = []
    for i in range(3, len(s)):
        for j in range(1, i):
            k = 1 if j == i - 1 else j + 1
            quads.append((ref, s[i], s[j], s[k]))
    return quads


def synthetic_obs(n_sites, n_frames=64, seed=37, uv_max=8.5e9, sigma_scale=1.0):
    """Synthetic observation table of the reference's layout: rows ordered
    (frame, baseline i<j); uv ~ U(-uv_max, uv_max); sigma ~ sigma_scale * U(0.1, 1.1).
    Returns dict(times, t1, t2, uv[T,F,2], sigma[T*F], tri, quad)."""
    rng = np.random.RandomState(seed)
    sites = ["S%d" % i for i in range(n_sites)]
    pairs = [(sites[i], sites[j]) for i in range(n_sites) for j in range(i + 1, n_sites)]
    F = len(pairs)
    uv = rng.uniform(-uv_max, uv_max, (n_frames, F, 2))
    sigma = rng.uniform(0.1, 1.1, n_frames * F) * sigma_scale
    times = 12.5 + np.arange(n_frames) * (102.0 / 3600.0)
    rows_time = np.repeat(times, F)
    t1 = [p[0] for p in pairs] * n_frames
    t2 = [p[1] for p in pairs] * n_frames
    tri = [(tm,) + t for tm in times for t in minimal_triangles(sites)]
    quad = [(tm,) + q for tm in times for q in minimal_quadrangles(sites)]
    return dict(times=rows_time, t1=t1, t2=t2, uv=uv, sigma=sigma, tri=tri, quad=quad,
                frame_times=times, sites=sites)


def ring_video(seed, n_frames=64, n=256, batch=1):
    """A compact, structured test video in [-1, 1] (float32, quantised to multiples of 2^-12 so
    that it is reproduced bit for bit wherever it is regenerated): a ring of ~40 uas diameter
    (80 px at the 0.5 uas pixel of blackhole.py:190) with a rotating brightness asymmetry plus a
    few Gaussian knots per frame -- the kind of image whose visibilities on EHT baselines are a
    sizeable fraction of the total flux, i.e. a WELL-CONDITIONED input for the closure quantities
    (uniform random pixels give |v| / sum|a x| ~ 1e-4 on the long baselines)."""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:n, 0:n].astype(np.float64)
    out = np.empty((batch, n_frames, 1, n, n), np.float32)
    for b in range(batch):
        ph0 = rng.uniform(0, 2 * np.pi)
        knots = rng.uniform(0.25, 0.75, (4, 2)) * n
        kamp = rng.uniform(0.2, 0.6, 4)
        ksig = rng.uniform(3.0, 9.0, 4) * n / 256.0
        for t in range(n_frames):
            cx = n / 2 + 3.0 * np.sin(0.11 * t + b) * n / 256.0
            cy = n / 2 + 3.0 * np.cos(0.07 * t + 2 * b) * n / 256.0
            r = np.hypot(xx - cx, yy - cy)
            th = np.arctan2(yy - cy, xx - cx)
            rad, wid = (40.0 + 2.0 * np.sin(0.05 * t)) * n / 256.0, 7.0 * n / 256.0
            img = np.exp(-0.5 * ((r - rad) / wid) ** 2) * (1.0 + 0.6 * np.cos(th - ph0 - 0.09 * t))
            for (ky, kx), a, s in zip(knots, kamp, ksig):
                img += a * np.exp(-0.5 * (((xx - kx - 0.2 * t) ** 2 + (yy - ky + 0.1 * t) ** 2) / s ** 2))
            img /= img.max()
            out[b, t, 0] = (np.round((2.0 * img - 1.0) * 4096.0) / 4096.0).astype(np.float32)
    return np.clip(out, -1.0, 1.0)
