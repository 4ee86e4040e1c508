#!/usr/bin/env python3
"""Full-size (64 x 256 x 256, 8 stations, F = 28) black-hole fixtures from the REFERENCE's own code
(build container only; same stubs and import route as oracle/gen_golden.py):

  bh_full_wc.npz     a WELL-CONDITIONED input (oracle/eht_synth.ring_video: min |v| / sum|a x| >= 1e-3
                     over all 1792 visibilities), so that the 1e-4 bar can be asserted outright on
                     visibilities, loss and gradient at the BASELINE shape;
  bh_full_seeds.npz  the ill-conditioned family (uniform random pixels) over several seeds: per seed
                     the reference's fp32 outputs and the fp64 evaluation of the same formulas, so
                     that each kernel path's error is a distribution and not one draw.

Run:  python oracle/gen_golden_full.py
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle import eht_synth, gen_golden as gg                    # noqa: E402
from oracle.bh_oracle import BlackholeOracle, Obs                 # noqa: E402

SHIPPED = dict(w_vis=0, w_amp=0, w_cphase=1, w_logcamp=1, w_flux=0.26, loss_normalize=True)
SUB = 97                       # gradient subsample stride (as in bh_full_n8)
OBS_SEED, SIGMA_SCALE = 37, 7e-3
WC_SEED_GT, WC_SEED_X = 201, 203
SEEDS = (101, 102, 103, 104)


def build(fo):
    syn = eht_synth.synthetic_obs(8, 64, seed=OBS_SEED, sigma_scale=SIGMA_SCALE)
    op = gg.ref_blackhole(fo, syn, 256, ref_multiplier=0.001, **SHIPPED)
    orc = gg.oracle_from_ref(op, SHIPPED)
    orc64 = BlackholeOracle(op.A_vis.to(torch.complex128), op.sigma.double(), op.cp_index,
                            op.cp_conjugate, op.camp_index, op.camp_conjugate, op.T, op.H, op.W, **SHIPPED)
    return syn, op, orc, orc64


def to64(y):
    return Obs(*[f.to(torch.complex128) if torch.is_complex(f) else f.double() for f in y.to_list()])


def evaluate(op, orc, orc64, x, y, tag, out, what):
    """reference fp32 + fp64 evaluation for one input; oracle == reference bit for bit."""
    yo, y64 = Obs(*y.to_list()), to64(y)
    o_ref, o_orc = op(x), orc(x)
    for n, a, b in zip(Obs._fields, o_ref.to_list(), o_orc):
        gg.close(a, b, what="%s obs.%s" % (what, n))
    loss = op.loss(x, y)
    gg.close(loss, orc.loss(x, yo), what=what + " loss")
    grad, lsum = op.gradient(x, y, return_loss=True)
    g2, l2 = orc.gradient(x, yo, return_loss=True)
    gg.close(grad, g2, what=what + " gradient")
    gg.close(lsum, l2, what=what + " loss sum")
    o64 = orc64(x.double())
    g64, l64 = orc64.gradient(x.double(), y64, return_loss=True)
    out.update(gg.obs_arrays(tag + "_obs_", o_ref))
    out.update(gg.obs_arrays(tag + "_obs64_", o64))
    out.update({tag + "_loss": loss, tag + "_loss_sum": lsum, tag + "_loss64": orc64.loss(x.double(), y64),
                tag + "_grad_sub": grad.flatten()[::SUB], tag + "_grad64_sub": g64.flatten()[::SUB],
                tag + "_grad_frame_sum": grad.flatten(2).sum(-1),
                tag + "_grad64_frame_sum": g64.flatten(2).sum(-1),
                tag + "_grad_abs_sum": grad.abs().sum()})
    return o64


def main():
    gg.install_stubs()
    fo, _, _, _ = gg.import_reference()
    torch.set_num_threads(8)
    syn, op, orc, orc64 = build(fo)
    common = dict(obs_seed=OBS_SEED, sigma_scale=SIGMA_SCALE, imsize=256, uv=syn["uv"],
                  cp_index=op.cp_index, cp_conjugate=op.cp_conjugate, camp_index=op.camp_index,
                  camp_conjugate=op.camp_conjugate, sigma=op.sigma,
                  weights=np.array([0, 0, 1, 1, 0.26], dtype=np.float64), loss_normalize=1)

    # ---- well-conditioned fixture
    x_gt = torch.from_numpy(eht_synth.ring_video(WC_SEED_GT))
    x = torch.from_numpy(eht_synth.ring_video(WC_SEED_X))
    y = op(x_gt)
    out = dict(common, seed_gt=WC_SEED_GT, seed_x=WC_SEED_X,
               x_checksum=float(x.double().sum()), x_gt_checksum=float(x_gt.double().sum()))
    out.update(gg.obs_arrays("y_", y))
    o64 = evaluate(op, orc, orc64, x, y, "x", out, "bh_full_wc")
    A = op.A_vis[0].to(torch.complex128)
    denom = (A.abs() * (x.double() * 0.5 + 0.5).reshape(64, 1, -1)).sum(-1).flatten()
    cond = (o64.vis[0].abs() / denom).min().item()
    print("well-conditioned case: min |v| / sum|a x| = %.3e" % cond)
    if cond < 1e-3:
        raise SystemExit("the well-conditioned fixture is not well conditioned")
    out["cond_min_ratio"] = cond
    gg.npz("bh_full_wc", **out)

    # ---- the ill-conditioned family over several seeds
    out = dict(common, seeds=np.array(SEEDS))
    for s in SEEDS:
        g = torch.Generator().manual_seed(s)
        x_gt = torch.rand(1, 64, 1, 256, 256, generator=g) * 2 - 1
        x = torch.rand(1, 64, 1, 256, 256, generator=g) * 2 - 1
        y = op(x_gt)
        tag = "s%d" % s
        out.update(gg.obs_arrays(tag + "_y_", y))
        o64 = evaluate(op, orc, orc64, x, y, tag, out, "bh_full_seeds/%d" % s)
        denom = (A.abs() * (x.double() * 0.5 + 0.5).reshape(64, 1, -1)).sum(-1).flatten()
        out[tag + "_cond_min_ratio"] = (o64.vis[0].abs() / denom).min().item()
        print("seed %d: min |v| / sum|a x| = %.3e" % (s, out[tag + "_cond_min_ratio"]))
    gg.npz("bh_full_seeds", **out)
    print("full-size fixtures written; oracle == reference on every compared value")


if __name__ == "__main__":
    main()
