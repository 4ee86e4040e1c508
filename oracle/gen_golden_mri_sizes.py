#!/usr/bin/env python3
"""Generate tests/golden/mri_sizes.npz from the REFERENCE's own DynamicMRI at resolutions other than the shipped
192 (build container only; run: python oracle/gen_golden_mri_sizes.py).  Same mechanism as oracle/gen_golden.py:
the reference's modules are imported from /root/reference behind stubs for the absent third-party names, every
fixture value is the reference's own arithmetic (forward_operator/dynamic_mri.py:9-48 with base.py:36-49), and
the oracle restatement is compared on the way (abort on mismatch).  These fixtures pin the any-size kernels
(step_amd/csrc/mri_any.hip) where the tuned 192 x 192 kernels do not apply: an even power of two, a 3- and
5-smooth size, an odd size (fftshift by (n - 1) / 2) and a prime."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import gen_golden as gg                # noqa: E402
from oracle.mri_oracle import MRIOracle            # noqa: E402

CASES = [  # (resolution, mode, acc, acs, frames, B, noise)
    (128, "random_split_mask", 4, 12, 6, 2, 0.05),
    (160, "same_mask", 6, 16, 3, 2, 0.0),
    (75, "split_mask", 3, 7, 4, 2, 0.05),
    (97, "same_mask", 2, 5, 2, 1, 0.05),
    (256, "same_mask", 8, 24, 2, 1, 0.05),
]


def main():
    gg.install_stubs()
    fo, *_ = gg.import_reference()
    out = {"cases": np.array([[c[0], c[2], c[3], c[4], c[5]] for c in CASES], np.int64),
           "modes": np.array([c[1] for c in CASES]), "noise": np.array([c[6] for c in CASES], np.float64)}
    for i, (res, mode, acc, acs, fr, B, noise) in enumerate(CASES):
        op = fo.get_operator("mri", acc_factor=acc, mode=mode, resolution=res, frames=fr, num_acs_lines=acs, sigma=noise)
        orc = MRIOracle(acc, mode, res, fr, acs, sigma=noise)
        gg.close(op.mask, orc.mask, what="mask %d" % res)
        rng = np.random.RandomState(1000 + res)
        shape = (B, fr, 2, res, res)
        x_gt, x, nz = (torch.from_numpy(rng.randn(*shape).astype(np.float32)) for _ in range(3))
        y = op(x_gt) + noise * nz
        fwd = op(x)
        grad, lsum = op.gradient(x, y, return_loss=True)
        g2, l2 = orc.gradient(x, y, return_loss=True)
        gg.close(fwd, orc(x), what="forward %d" % res)
        gg.close(grad, g2, what="gradient %d" % res)
        gg.close(lsum, l2, what="loss %d" % res)
        k = "c%d_" % i
        out.update({k + "mask": op.mask[0, :, 0, :, 0].to(torch.uint8), k + "fwd_sub": fwd.flatten()[::7],
                    k + "fwd_frame0": fwd[0, 0], k + "grad_sub": grad.flatten()[::7], k + "grad_frame0": grad[0, 0],
                    k + "grad_sq_sum": (grad.double() ** 2).flatten(1).sum(-1), k + "loss": op.loss(x, y),
                    k + "loss_sum": lsum})
    gg.npz("mri_sizes", **out)


if __name__ == "__main__":
    main()
