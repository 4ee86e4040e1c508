"""CPU emulation of the summation order of k_dft_fwd_sep (fp32 MFMA path) on the ill-conditioned
seeds of tests/golden/bh_full_seeds.npz: which level of the sum carries the visibility error, and
what each candidate order would give.  fp32 arithmetic emulated with numpy float32 (every MFMA
k-step = two fused multiply-adds in sequence; the hardware's internal order is not documented)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from step_amd import eht_lite
g = np.load(os.path.join(os.path.dirname(__file__), "../../tests/golden/bh_full_seeds.npz"))
tab = eht_lite.synthetic_table(8, 64, seed=int(g["obs_seed"]), sigma_scale=float(g["sigma_scale"]))
ey, ex = eht_lite.ft_factors(np.asarray(tab["uv"], np.float64), 256, 256, eht_lite.PSIZE)   # [T,F,H], [T,F,W]
T, F, H = ey.shape
W = ex.shape[2]
f32 = np.float32
EY = np.concatenate([ey.real, ey.imag], 1).astype(f32)          # [T, 2F, H]
EXr, EXi = ex.real.astype(f32), ex.imag.astype(f32)

def fma(a, b, c):   # fp32 fused multiply-add through float64 (exact product, one rounding)
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)

def column_dft(x01, chunk, flush):
    """Z[t, m, x] = sum_y EY[t, m, y] x01[t, y, x]; MFMA chains of `chunk` rows, flushed by `flush`."""
    tot = np.zeros((T, 2 * F, W), f32)
    lo = np.zeros_like(tot); comp = np.zeros_like(tot); n_lo = 0
    for y0 in range(0, H, chunk):
        acc = np.zeros_like(tot)
        for y in range(y0, y0 + chunk):
            acc = fma(EY[:, :, y, None], x01[:, None, y, :], acc)
        if flush == "seq":
            tot = tot + acc
        elif flush == "two":                       # a second-level accumulator folded every 4 chunks
            lo = lo + acc; n_lo += 1
            if n_lo == 4:
                tot = tot + lo; lo = np.zeros_like(tot); n_lo = 0
        elif flush == "kahan":                     # TwoSum-compensated running total
            s = tot + acc
            bb = s - tot
            comp = comp + ((tot - (s - bb)) + (acc - bb))
            tot = s
    if flush == "kahan":
        tot = tot + comp
    return tot

def reduce_x(Z, how):
    Zr, Zi = Z[:, :F], Z[:, F:]
    pr = fma(EXr, Zr, -(EXi * Zi)); pi = fma(EXr, Zi, EXi * Zr)      # [T, F, W]
    out = []
    for p in (pr, pi):
        if how == "seq":
            s = np.zeros((T, F, W // 32), f32)
            ps = p.reshape(T, F, W // 32, 32)
            for j in range(32):
                s = s + ps[..., j]
            v = np.zeros((T, F), f32)
            for w in range(W // 32):
                v = v + s[..., w]
        else:                                       # pairwise
            v = p
            while v.shape[-1] > 1:
                v = v[..., 0::2] + v[..., 1::2]
            v = v[..., 0]
        out.append(v)
    return out[0].astype(np.float64) + 1j * out[1].astype(np.float64)

for s in [int(v) for v in g["seeds"]][2:]:
    gen = torch.Generator().manual_seed(s)
    torch.rand(1, 64, 1, 256, 256, generator=gen)
    x = (torch.rand(1, 64, 1, 256, 256, generator=gen) * 2 - 1).numpy()[0, :, 0]
    x01 = (x.astype(f32) * f32(0.5) + f32(0.5)).astype(f32)
    v64 = np.einsum("tfy,tyx,tfx->tf", ey, x01.astype(np.float64), ex)
    scale = np.einsum("tfy,tyx,tfx->tf", np.abs(ey), np.abs(x01.astype(np.float64)), np.abs(ex))
    worst = np.unravel_index(np.argmin(np.abs(v64) / scale), v64.shape)
    print("seed %d: min |v|/sum|ax| = %.1e at %s" % (s, (np.abs(v64) / scale)[worst], worst))
    for name, chunk, flush, how in (("now: 16-row chains, sequential flush, sequential x", 16, "seq", "seq"),
                                    ("16-row chains, sequential flush, pairwise x", 16, "seq", "pair"),
                                    ("16-row chains, 2-level flush, pairwise x", 16, "two", "pair"),
                                    ("16-row chains, compensated flush, pairwise x", 16, "kahan", "pair"),
                                    ("8-row chains, compensated flush, pairwise x", 8, "kahan", "pair"),
                                    ("32-row chains, compensated flush, pairwise x", 32, "kahan", "pair")):
        v = reduce_x(column_dft(x01, chunk, flush), how)
        e = np.abs(v - v64)
        inc = np.abs(v64) / scale < 1e-2            # incoherent sums: their error is the summation noise
        print("   %-52s rms err/sum|ax| over the %d incoherent vis %.2e   rel err of the worst-conditioned vis %.2e"
              % (name, inc.sum(), np.sqrt(np.mean((e / scale)[inc] ** 2)), e[worst] / abs(v64[worst])))
