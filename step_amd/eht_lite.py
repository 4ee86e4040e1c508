"""ehtim-free set-up of the black-hole operator (host side, numpy fp64, runs once).

PARITY UNPINNED: the reference obtains all of this from ``ehtim==1.2.10``
(forward_operator/blackhole.py:188-256: ``eh.array.load_txt``, ``Array.obsdata``,
``obsh.ftmatrix`` with ``trianglePulse2D``, ``Obsdata.c_phases`` / ``c_amplitudes``), which is
neither vendored in the reference nor installed in the build image.  What follows restates
the published behaviour of those routines; values derived from it cannot be checked against
ehtim here.  Everything *downstream* (the index mapping of blackhole.py:235-264 and all
operator arithmetic) is pinned by golden fixtures.

An "observation table" is a dict with
    frame_times [T] float64, times [T*F] float64, t1 / t2 [T*F] station names,
    uv [T, F, 2] float64 (wavelengths), sigma [T*F] float64 (Jy),
    tri  list of (time, a, b, c), quad list of (time, a, b, c, d)
with rows ordered (frame, baseline) as the reference assumes (blackhole.py:223-241).
"""
import numpy as np

PSIZE = 2.4240684055470684e-12          # blackhole.py:190 (rad / pixel)
C_LIGHT = 299792458.0
DEG = np.pi / 180.0
HOUR = 15.0 * DEG

# observation constants of process_obs, blackhole.py:190-210
EHT_DEFAULTS = dict(ra=17.761120, dec=-29.007797, rf=230e9, bw=2e9, mjd=60775, tint=102.0,
                    tadv=102.0, tau=0.1, elevmin=10.0, elevmax=85.0, tstart=12.5, tstop=14.3)


# ----------------------------------------------------------------------------- ftmatrix factors
def tri_pulse_ft(omega, pdim):
    """Fourier transform of the 1-D triangle pulse of width 2*pdim at angular frequency omega."""
    omega = np.asarray(omega, dtype=np.float64)
    safe = np.where(omega == 0, 1.0, omega)
    val = 4.0 / (pdim ** 2 * safe ** 2) * np.sin(pdim * safe / 2.0) ** 2
    return np.where(omega == 0, 1.0, val)


def pixel_coords(n, pdim):
    return np.arange(0, -n, -1) * pdim + (pdim * n) / 2.0 - pdim / 2.0


def ft_factors(uv, H, W, pdim=PSIZE):
    """Separable factors of ftmatrix(pdim, W, H, uv, pulse=trianglePulse2D):
    A[f, y*W+x] = ey[f, y] * ex[f, x].  uv [..., F, 2] -> ey [..., F, H], ex [..., F, W] complex128
    (the pulse is folded into ey)."""
    uv = np.asarray(uv, dtype=np.float64)
    u, v = uv[..., 0], uv[..., 1]
    pulse = tri_pulse_ft(2 * np.pi * u, pdim) * tri_pulse_ft(2 * np.pi * v, pdim)
    ey = pulse[..., None] * np.exp(2j * np.pi * v[..., None] * pixel_coords(H, pdim))
    ex = np.exp(2j * np.pi * u[..., None] * pixel_coords(W, pdim))
    return ey, ex


def dense_from_factors(ey, ex):
    """[..., F, H], [..., F, W] -> [..., F, H*W] (what the reference stores as A_vis)."""
    return (ey[..., :, None] * ex[..., None, :]).reshape(*ey.shape[:-1], -1)


# ----------------------------------------------------------------------------- minimal closure sets
def minimal_triangles(sites):
    """(N-1)(N-2)/2 triangles through the reference station sites[0] (lowest SEFD first)."""
    ref, rest = sites[0], list(sites[1:])
    return [(ref, rest[i], rest[j]) for i in range(len(rest)) for j in range(i + 1, len(rest))]


def minimal_quadrangles(sites):
    """N(N-3)/2 quadrangles, ordered (12)(34)/(14)(23)."""
    s = list(sites)
    quads = []
    for i in range(3, len(s)):
        for j in range(1, i):
            k = 1 if j == i - 1 else j + 1
            quads.append((s[0], s[i], s[j], s[k]))
    return quads


# ----------------------------------------------------------------------------- synthetic table
def synthetic_table(n_sites=8, n_frames=64, seed=37, uv_max=8.5e9, sigma_scale=1.0):
    """Random uv coverage of the reference's shape: every one of the
    n_sites(n_sites-1)/2 baselines present in every frame, uv ~ U(-uv_max, uv_max),
    sigma ~ sigma_scale * U(0.1, 1.1)."""
    rng = np.random.RandomState(seed)
    sites = ["S%d" % i for i in range(n_sites)]
    pairs = [(sites[i], sites[j]) for i in range(n_sites) for j in range(i + 1, n_sites)]
    F = len(pairs)
    uv = rng.uniform(-uv_max, uv_max, (n_frames, F, 2))
    sigma = rng.uniform(0.1, 1.1, n_frames * F) * sigma_scale
    ft = 12.5 + np.arange(n_frames) * (102.0 / 3600.0)
    return dict(frame_times=ft, times=np.repeat(ft, F), t1=[p[0] for p in pairs] * n_frames,
                t2=[p[1] for p in pairs] * n_frames, uv=uv, sigma=sigma,
                tri=[(tm,) + t for tm in ft for t in minimal_triangles(sites)],
                quad=[(tm,) + q for tm in ft for q in minimal_quadrangles(sites)], sites=sites)


# ----------------------------------------------------------------------------- array file + earth rotation
def load_array_txt(path):
    """Station table: NAME X Y Z SEFDR SEFDL ... (configs/task/EHT2017.txt)."""
    sites = []
    with open(path) as fh:
        for line in fh:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            p = line.split()
            sites.append(dict(name=p[0], xyz=np.array([float(p[1]), float(p[2]), float(p[3])]),
                              sefdr=float(p[4]), sefdl=float(p[5])))
    return sites


def gmst_hours(mjd):
    """Greenwich mean sidereal time (hours) at fractional MJD (IAU 1982 linear term)."""
    d = np.asarray(mjd, dtype=np.float64) - 51544.5
    return np.mod(18.697374558 + 24.06570982441908 * d, 24.0)


def observe_array(sites, ra, dec, rf, bw, tint, tadv, tstart, tstop, mjd, tau=0.1,
                  elevmin=10.0, elevmax=85.0):
    """Earth-rotation synthesis: for every time step the baselines whose two stations see the
    source between elevmin and elevmax; thermal sigma = sqrt(SEFD1 SEFD2 / (2 bw tint)) / 0.88.
    Raises ValueError if the number of baselines changes between frames (the reference
    np.stack's per-frame matrices, blackhole.py:230, and fails the same way)."""
    if tstop < tstart:
        tstop += 24.0
    times = np.arange(tstart, tstop, tadv / 3600.0)
    lam = C_LIGHT / rf
    src = np.array([np.cos(dec * DEG), 0.0, np.sin(dec * DEG)])
    proj_u = np.array([0.0, 1.0, 0.0])
    proj_v = np.array([-np.sin(dec * DEG), 0.0, np.cos(dec * DEG)])
    order = sorted(range(len(sites)), key=lambda i: sites[i]["sefdr"] + sites[i]["sefdl"])
    rows, per_frame = [], []
    for tm in times:
        th = np.mod((gmst_hours(mjd + tm / 24.0) - ra) * HOUR, 2 * np.pi)
        rot = np.array([[np.cos(th), -np.sin(th), 0.0], [np.sin(th), np.cos(th), 0.0], [0, 0, 1.0]])
        pos = {s["name"]: rot @ s["xyz"] for s in sites}
        up = {}
        for s in sites:
            p = pos[s["name"]]
            el = np.arcsin(np.dot(p, src) / np.linalg.norm(p)) / DEG
            up[s["name"]] = elevmin <= el <= elevmax
        frame = []
        for i in range(len(sites)):
            for j in range(i + 1, len(sites)):
                a, b = sites[i], sites[j]
                if not (up[a["name"]] and up[b["name"]]):
                    continue
                bl = (pos[a["name"]] - pos[b["name"]]) / lam
                sig_r = np.sqrt(a["sefdr"] * b["sefdr"] / (2 * bw * tint)) / 0.88
                sig_l = np.sqrt(a["sefdl"] * b["sefdl"] / (2 * bw * tint)) / 0.88
                frame.append((tm, a["name"], b["name"], float(bl @ proj_u), float(bl @ proj_v),
                              0.5 * np.sqrt(sig_r ** 2 + sig_l ** 2)))
        per_frame.append(frame)
        rows.extend(frame)
    counts = {len(f) for f in per_frame}
    if len(counts) != 1 or 0 in counts:
        raise ValueError("observe_array: the number of baselines per frame is not constant "
                         "(%s); the operator needs a fixed F" % sorted(counts))
    F = counts.pop()
    tri, quad = [], []
    for tm, frame in zip(times, per_frame):
        present = {r[1] for r in frame} | {r[2] for r in frame}
        seen = [sites[i]["name"] for i in order if sites[i]["name"] in present]
        tri += [(tm,) + t for t in minimal_triangles(seen)]
        quad += [(tm,) + q for q in minimal_quadrangles(seen)]
    return dict(frame_times=times, times=np.array([r[0] for r in rows]),
                t1=[r[1] for r in rows], t2=[r[2] for r in rows],
                uv=np.array([[r[3], r[4]] for r in rows]).reshape(len(times), F, 2),
                sigma=np.array([r[5] for r in rows]), tri=tri, quad=quad,
                sites=[s["name"] for s in sites])


def eht_table(array_path, **overrides):
    """Observation table for an array file with the constants of process_obs."""
    kw = dict(EHT_DEFAULTS)
    kw.update(overrides)
    return observe_array(load_array_txt(array_path), **kw)
