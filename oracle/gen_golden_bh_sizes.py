#!/usr/bin/env python3
"""Black-hole fixtures from the REFERENCE's own code at image sizes other than the 256 x 256 of BASELINE (build
container only; same stubs and import route as oracle/gen_golden.py; run: python oracle/gen_golden_bh_sizes.py):

  bh_sizes.npz   64 x N x N for N = 64 and 128 (the other two instantiations of the folded 3 x bf16 kernels) and
                 N = 96 and 160 (sizes only the fp32 separable kernels are built for), 6 stations (F = 15), on
                 a well-conditioned ring video scaled to the image; per size the reference's fp32 observation,
                 loss and gradient and the fp64 evaluation of the same formulas.

The oracle is compared on the way and the script aborts on a mismatch."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle import eht_synth, gen_golden as gg, gen_golden_full as gf      # noqa: E402
from oracle.bh_oracle import BlackholeOracle                               # noqa: E402

SIZES = (64, 128, 96, 160)
N_SITES, OBS_SEED = 6, 41


def main():
    gg.install_stubs()
    fo, _, _, _ = gg.import_reference()
    torch.set_num_threads(8)
    out = dict(sizes=np.array(SIZES), n_sites=N_SITES, obs_seed=OBS_SEED,
               weights=np.array([0, 0, 1, 1, 0.26], dtype=np.float64), loss_normalize=1)
    for n in SIZES:
        sigma_scale = 7e-3 * n / 256.0
        syn = eht_synth.synthetic_obs(N_SITES, 64, seed=OBS_SEED, sigma_scale=sigma_scale)
        op = gg.ref_blackhole(fo, syn, n, ref_multiplier=0.001, **gf.SHIPPED)
        orc = gg.oracle_from_ref(op, gf.SHIPPED)
        orc64 = BlackholeOracle(op.A_vis.to(torch.complex128), op.sigma.double(), op.cp_index, op.cp_conjugate,
                                op.camp_index, op.camp_conjugate, op.T, op.H, op.W, **gf.SHIPPED)
        x_gt = torch.from_numpy(eht_synth.ring_video(300 + n, n=n))
        x = torch.from_numpy(eht_synth.ring_video(400 + n, n=n))
        y = op(x_gt)
        tag = "n%d" % n
        out.update({tag + "_sigma_scale": sigma_scale, tag + "_uv": syn["uv"], tag + "_cp_index": op.cp_index,
                    tag + "_camp_index": op.camp_index, tag + "_x_checksum": float(x.double().sum())})
        out.update(gg.obs_arrays(tag + "_y_", y))
        o64 = gf.evaluate(op, orc, orc64, x, y, tag, out, "bh_sizes/%d" % n)
        A = op.A_vis[0].to(torch.complex128)
        denom = (A.abs() * (x.double() * 0.5 + 0.5).reshape(64, 1, -1)).sum(-1).flatten()
        out[tag + "_cond_min_ratio"] = (o64.vis[0].abs() / denom).min().item()
        print("N = %d: min |v| / sum|a x| = %.3e" % (n, out[tag + "_cond_min_ratio"]))
    gg.npz("bh_sizes", **out)


if __name__ == "__main__":
    main()
