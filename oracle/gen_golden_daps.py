#!/usr/bin/env python3
"""Golden fixture for the annealing driver: tests/golden/daps_bh_small.npz from the REFERENCE's own
``LatentDAPS.sample`` (sampler/daps.py:254-382), build container only.

Run:  python oracle/gen_golden_daps.py        (after oracle/gen_golden.py; needs /root/reference)

Same machinery as gen_golden.py (stubs for the absent third-party names only).  The reference
integrates the PF-ODE with ``torchdiffeq.odeint(..., method='euler')`` (cores/scheduler.py:402);
torchdiffeq 0.2.x is not installed, so its stub here provides the published behaviour of the
fixed-grid Euler solver -- the grid is the requested time vector, y_{i+1} = y_i + (t_{i+1} - t_i)
f(t_i, y_i), all grid points returned -- and that one routine is UNPINNED.  Everything else that
produces the fixture (annealing schedule, PF-ODE derivative, MCMC loop, re-noising, decode calls,
the optimize branch) is the reference's own code on the operator of the bh_small fixture.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G                                   # noqa: E402
from oracle import eht_synth                             # noqa: E402
from tests.toy_model import ToyModel                     # noqa: E402


def euler_odeint(func, y0, t, rtol=None, atol=None, method="euler", **kw):
    assert method == "euler"
    ys = [y0]
    for i in range(len(t) - 1):
        ys.append(ys[-1] + (t[i + 1] - t[i]) * func(t[i], ys[-1]))
    return torch.stack(ys)


def main():
    G.install_stubs()
    sys.modules["torchdiffeq"].odeint = euler_odeint
    fo, LatentWrapper, MCMCSampler, get_sched = G.import_reference()
    from sampler.daps import LatentDAPS
    torch.set_num_threads(8)
    shipped = dict(w_vis=0, w_amp=0, w_cphase=1, w_logcamp=1, w_flux=0.26, loss_normalize=True)
    # the operator and measurement of bh_small (same arguments as gen_golden.main)
    syn = eht_synth.synthetic_obs(4, 64, seed=37, sigma_scale=0.5e-3)
    op = G.ref_blackhole(fo, syn, 16, ref_multiplier=0.001, **shipped)
    g = torch.Generator().manual_seed(37)
    x_gt = torch.rand(2, 64, 1, 16, 16, generator=g) * 2 - 1
    y = op(x_gt)
    gold = np.load(os.path.join(G.GOLD, "bh_small.npz"))
    assert np.array_equal(gold["y_cphase"], y.cphase.numpy()), "run gen_golden.py first: bh_small differs"

    model = ToyModel((64, 1, 8, 8), factor=2, gain=0.02)
    ann = dict(name="edm", num_steps=5, sigma_max=10, sigma_min=1e-2, timestep="poly-7")
    dif = dict(name="edm", num_steps=4, sigma_min=1e-5, timestep="poly-7")
    out = dict(ann_num_steps=5, ann_sigma_max=10.0, ann_sigma_min=1e-2, dif_num_steps=4,
               dif_sigma_min=1e-5, toy_gain=0.02)
    cases = {
        "latent": (dict(num_steps=3, lr=1.2e-5, tau=0.021, lr_min_ratio=0.015, prior_solver="exact",
                        mc_algo="hmc", momentum=0.0), True),
        "latent_mom": (dict(num_steps=3, lr=1.2e-4, tau=0.1, lr_min_ratio=0.011, prior_solver="gaussian",
                            mc_algo="hmc", momentum=0.83), True),
        "pixel": (dict(num_steps=3, lr=2e-7, tau=1.0), False),
    }
    for name, (mc, wrap) in cases.items():
        torch.manual_seed(41)
        s = LatentDAPS(dict(ann), dict(dif), dict(mc), latent_wrapper=wrap)
        z_start = torch.randn(2, 64, 1, 8, 8) * s.annealing_scheduler.sigma_max
        with G.NoiseTape() as tape:
            res = s.sample(model, z_start, op, y)
        assert torch.isfinite(res).all()
        out[name + "_z_start"] = z_start
        out[name + "_noise"] = torch.stack(tape.items)
        out[name + "_result"] = res
        for k, v in mc.items():
            if not isinstance(v, str):
                out["%s_mc_%s" % (name, k)] = v
    G.npz("daps_bh_small", **out)

    # the MRI task's shipped sampler algorithm (hmc, momentum 0.83, exact prior, lr 1.2e-3;
    # configs/task/mri.yaml:48-56) on the reference's own 'mri' operator, two annealing steps, one
    # chain.  tau is 30 instead of 0.1: with the toy decoder the shipped value makes lr/tau^2 times
    # the loss curvature ~1e4, a chaotic map on which fp32 rounding differences grow to 1e-3 in
    # three steps -- no yardstick for parity
    mop = fo.get_operator("mri", acc_factor=6, mode="same_mask", resolution=192, frames=12,
                          num_acs_lines=24)
    g = torch.Generator().manual_seed(38)
    ym = mop(torch.randn(1, 12, 2, 192, 192, generator=g))
    mmodel = ToyModel((12, 2, 48, 48), factor=4, gain=0.5)
    mc = dict(num_steps=3, lr=1.2e-3, tau=30.0, lr_min_ratio=0.011, prior_solver="exact",
              mc_algo="hmc", momentum=0.83)
    torch.manual_seed(43)
    s = LatentDAPS(dict(name="edm", num_steps=2, sigma_max=10, sigma_min=1e-2, timestep="poly-7"),
                   dict(name="edm", num_steps=3, sigma_min=1e-5, timestep="poly-7"), dict(mc),
                   latent_wrapper=True)
    z_start = torch.randn(1, 12, 2, 48, 48) * s.annealing_scheduler.sigma_max
    with G.NoiseTape() as tape:
        res = s.sample(mmodel, z_start, mop, ym)
    assert torch.isfinite(res).all()
    # the measurement is stored as its kept k-space rows only (the masked-out rows are exactly 0)
    rows = torch.nonzero(mop.mask.flatten()).flatten()
    assert torch.count_nonzero(ym) == torch.count_nonzero(ym[:, :, :, rows])
    G.npz("daps_mri_small", z_start=z_start, noise=torch.stack(tape.items), result=res,
          y_rows=ym[:, :, :, rows], rows=rows, toy_gain=0.5)


if __name__ == "__main__":
    main()
