"""CPU: the oracle restatement reproduces the golden fixtures (= outputs of the reference's own
code, oracle/gen_golden.py).  This is what pins the oracle; the GPU tests then use the oracle."""
import numpy as np
import pytest
import torch

from oracle.bh_oracle import BlackholeOracle, Obs, closure_tables
from oracle.mcmc_oracle import LatentWrapOracle, MCMCOracle, edm_sigmas, lr_at
from oracle.mri_oracle import MRIOracle
from tests.helpers import obs_from_golden, oracle_from_golden, t, weights_kw
from tests.toy_model import ToyModel

TIGHT = 2e-6      # same formulas, same dtype: only BLAS/vectorisation differences between CPUs


def close(a, b, tol=TIGHT):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    if torch.is_complex(a):
        a, b = torch.view_as_real(a), torch.view_as_real(b)
    assert ((a - b).abs().max() <= tol * b.abs().max().clamp_min(1e-30)), \
        ((a - b).abs().max().item(), b.abs().max().item())


@pytest.mark.parametrize("name", ["bh_small", "bh_allterms", "bh_allterms_norm"])
def test_blackhole_oracle_matches_reference_outputs(name, golden):
    g = golden(name)
    orc = oracle_from_golden(g)
    y = obs_from_golden(g, "y_")
    for tag, key in (("x", "x"), ("near", "x_near")):
        x = t(g[key])
        for f, got in zip(Obs._fields, orc(x)):
            close(got, g["%s_obs_%s" % (tag, f)])
        close(orc.loss(x, y), g[tag + "_loss"], 1e-5)
        grad, lsum = orc.gradient(x, y, return_loss=True)
        close(grad, g[tag + "_grad"], 1e-5)
        close(lsum, g[tag + "_loss_sum"], 1e-5)
        ev = orc.evaluate_chisq(x, y, normalize=False)
        for k in ev:
            close(ev[k], g["%s_eval_%s" % (tag, k)], 1e-5)
    model = ToyModel((64, 1, int(g["imsize"]) // 2, int(g["imsize"]) // 2), 2, float(g["toy_gain"]))
    zg, zl = LatentWrapOracle(orc, model).gradient(t(g["z"]), y, return_loss=True)
    close(zl, g["z_loss_sum"], 1e-5)
    assert (zg - t(g["z_grad"])).abs().max() <= 1e-4


@pytest.mark.parametrize("name", ["bh_small", "bh_allterms", "bh_allterms_norm", "bh_full_n8"])
def test_closure_index_mapping_is_bit_exact(name, golden):
    g = golden(name)
    tri = [(tm,) + tuple(s) for tm, s in zip(np.repeat(np.unique(g["row_time"]), len(g["tri_sites"]) // 64), g["tri_sites"])]
    quad = [(tm,) + tuple(s) for tm, s in zip(np.repeat(np.unique(g["row_time"]), len(g["quad_sites"]) // 64), g["quad_sites"])]
    cpi, cpc, cai, cac = closure_tables(g["row_time"], list(g["row_t1"]), list(g["row_t2"]), tri, quad)
    assert np.array_equal(np.array(cpi), g["cp_index"]) and np.array_equal(np.array(cpc), g["cp_conjugate"])
    assert np.array_equal(np.array(cai), g["camp_index"]) and np.array_equal(np.array(cac), g["camp_conjugate"])
    assert g["cp_conjugate"].sum() > 0          # the conjugated-leg branch is exercised


@pytest.mark.parametrize("mode", ["same_mask", "split_mask", "random_split_mask"])
def test_mri_oracle_matches_reference_outputs(mode, golden):
    g = golden("mri_" + mode)
    orc = MRIOracle(6, mode, 192, 12, 24, sigma=float(g["noise"]))
    assert np.array_equal(orc.mask[0, :, 0, :, 0].numpy().astype(np.uint8), g["mask"])
    rng = np.random.RandomState(int(g["seed"]))
    shape = (int(g["B"]), 12, 2, 192, 192)
    x_gt, x, nz = (torch.from_numpy(rng.randn(*shape).astype(np.float32)) for _ in range(3))
    y = orc(x_gt) + float(g["noise"]) * nz
    fwd = orc(x)
    close(fwd.flatten()[::11], g["fwd_sub"], 1e-5)
    close(fwd[0, 0], g["fwd_frame0"], 1e-5)
    grad, lsum = orc.gradient(x, y, return_loss=True)
    close(grad.flatten()[::11], g["grad_sub"], 1e-5)
    close(lsum, g["loss_sum"], 1e-5)
    close(orc.loss(x, y), g["loss"], 1e-5)


def mri_size_cases(g):
    """(index, resolution, mode, acc, acs, frames, B, noise) of tests/golden/mri_sizes.npz (written from the
    reference by oracle/gen_golden_mri_sizes.py)."""
    return [(i, int(c[0]), str(g["modes"][i]), int(c[1]), int(c[2]), int(c[3]), int(c[4]), float(g["noise"][i]))
            for i, c in enumerate(g["cases"])]


def mri_size_inputs(res, frames, B):
    rng = np.random.RandomState(1000 + res)
    shape = (B, frames, 2, res, res)
    return tuple(torch.from_numpy(rng.randn(*shape).astype(np.float32)) for _ in range(3))


def test_mri_oracle_matches_reference_outputs_at_other_resolutions(golden):
    g = golden("mri_sizes")
    for i, res, mode, acc, acs, fr, B, noise in mri_size_cases(g):
        orc = MRIOracle(acc, mode, res, fr, acs, sigma=noise)
        k = "c%d_" % i
        assert np.array_equal(orc.mask[0, :, 0, :, 0].numpy().astype(np.uint8), g[k + "mask"])
        x_gt, x, nz = mri_size_inputs(res, fr, B)
        y = orc(x_gt) + noise * nz
        fwd = orc(x)
        close(fwd.flatten()[::7], g[k + "fwd_sub"], 1e-5)
        close(fwd[0, 0], g[k + "fwd_frame0"], 1e-5)
        grad, lsum = orc.gradient(x, y, return_loss=True)
        close(grad.flatten()[::7], g[k + "grad_sub"], 1e-5)
        close(lsum, g[k + "loss_sum"], 1e-5)
        close(orc.loss(x, y), g[k + "loss"], 1e-5)


def test_mri_masks_all_configs(golden):
    from oracle.mri_oracle import build_mask
    d = golden("mri_masks")
    for key, want in d.items():
        mode, a, c, r, f = key.rsplit("_", 4)
        m = build_mask(int(a[1:]), mode, int(r[1:]), int(f[1:]), int(c[1:]))
        assert np.array_equal(m[0, :, 0, :, 0].numpy().astype(np.uint8), want), key


def test_mcmc_oracle_matches_reference_trajectories(golden):
    from tests.test_gpu_mcmc import CASES, Replay
    g, m = golden("bh_small"), golden("mcmc_bh_small")
    orc, y = oracle_from_golden(g), obs_from_golden(g, "y_")
    model = ToyModel((64, 1, 8, 8), factor=2, gain=1.5)
    for case, (cfg, space) in CASES.items():
        s = MCMCOracle(**cfg)
        if space == "latent":
            oper, start, xt = LatentWrapOracle(orc, model), t(m["z0hat"]), t(m["zt"])
        else:
            oper, start, xt = orc, t(m["x0hat"]), t(m[case + "_xt"])
        with Replay(list(t(m[case + "_noise"]))):
            out = s.sample(xt, model, start, oper, y, float(m["sigma"]), float(m["ratio"]), record=True)
        close(out, m[case + "_result"], 1e-4)
        close(torch.stack(s.trace), m[case + "_traj"], 1e-4)
        assert lr_at(cfg["lr"], cfg.get("lr_min_ratio", 0.01), float(m["ratio"])) == float(m[case + "_lr"])
    with Replay(list(t(m["mh_pixel_noise"])), list(t(m["mh_pixel_unif"]))):
        out = MCMCOracle(num_steps=6, lr=1e-7, tau=3.0, mc_algo="MH").sample(
            t(m["mh_pixel_xt"]), model, t(m["x0hat"]), orc, y, float(m["sigma"]), float(m["ratio"]))
    close(out, m["mh_pixel_result"], 1e-4)
    o = MCMCOracle(num_steps=5, lr=2e-7, tau=1.0).optimize(t(m["x0hat"]), orc, y, 0.7, 0.25)
    close(o, m["optimize_result"], 1e-4)


def test_edm_sigma_grid(golden):
    e = golden("edm_sigmas")
    assert torch.equal(edm_sigmas(25), t(e["s25"])) and torch.equal(edm_sigmas(20), t(e["s20"]))
    assert len(e["s25"]) == 26 and e["s25"][0] == 100 and e["s25"][-1] == 0


def test_latent_daps_oracle_matches_reference(golden):
    """The annealing driver restated (oracle.mcmc_oracle.LatentDAPSOracle) against the reference's own
    LatentDAPS.sample (tests/golden/daps_bh_small.npz, reference noise replayed)."""
    from oracle.mcmc_oracle import LatentDAPSOracle
    from tests.test_gpu_mcmc import DAPS_CASES, Replay
    g, d = golden("bh_small"), golden("daps_bh_small")
    orc, y = oracle_from_golden(g), obs_from_golden(g, "y_")
    model = ToyModel((64, 1, 8, 8), factor=2, gain=float(d["toy_gain"]))
    ann = dict(num_steps=int(d["ann_num_steps"]), sigma_max=float(d["ann_sigma_max"]),
               sigma_min=float(d["ann_sigma_min"]), timestep="poly-7")
    dif = dict(num_steps=int(d["dif_num_steps"]), sigma_min=float(d["dif_sigma_min"]), timestep="poly-7")
    for case, (mc, wrap) in DAPS_CASES.items():
        with Replay(list(t(d[case + "_noise"]))) as tape:
            out = LatentDAPSOracle(ann, dif, mc, latent_wrapper=wrap).sample(model, t(d[case + "_z_start"]), orc, y)
        assert not tape.n
        close(out, d[case + "_result"], 1e-4)


def test_latent_daps_oracle_matches_reference_mri(golden):
    from oracle.mcmc_oracle import LatentDAPSOracle
    from tests.test_gpu_mcmc import MRI_DAPS, Replay, mri_measurement
    d = golden("daps_mri_small")
    orc = MRIOracle(6, "same_mask", 192, 12, 24)
    model = ToyModel((12, 2, 48, 48), factor=4, gain=float(d["toy_gain"]))
    with Replay(list(t(d["noise"]))) as tape:
        out = LatentDAPSOracle(MRI_DAPS["ann"], MRI_DAPS["dif"], MRI_DAPS["mc"]).sample(
            model, t(d["z_start"]), orc, mri_measurement(d))
    assert not tape.n
    close(out, d["result"], 1e-4)


def test_multicoil_oracle_reduces_to_the_pinned_single_coil_oracle():
    """The multi-coil oracle has no reference counterpart; its C = 1, S = 1 case IS the pinned
    single-coil oracle, bit for bit, and its coil sum is explicit."""
    from oracle.mri_mc_oracle import MultiCoilMRIOracle
    sens = torch.ones(1, 192, 192, dtype=torch.complex64)
    mc, sc = MultiCoilMRIOracle(6, "same_mask", 192, 12, 24, sens), MRIOracle(6, "same_mask", 192, 12, 24)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 12, 2, 192, 192, generator=g)
    y = torch.randn(1, 12, 2, 192, 192, generator=g)
    assert torch.equal(mc(x)[:, :, 0], sc(x))
    assert torch.equal(mc.loss(x, y.unsqueeze(2)), sc.loss(x, y))
    s2 = torch.stack([sens[0], 2j * sens[0]])
    m2 = MultiCoilMRIOracle(6, "same_mask", 192, 12, 24, s2)
    want = sc.loss(x, y) + (sc(2 * torch.stack([-x[:, :, 1], x[:, :, 0]], 2)) - y).pow(2).flatten(1).sum(-1)
    close(m2.loss(x, torch.stack([y, y], 2)), want, 1e-6)


def test_blackhole_oracle_matches_reference_outputs_at_another_image_size(golden):
    """tests/golden/bh_sizes.npz, N = 64: the oracle rebuilt from the fixture's uv table reproduces the reference's
    observation, loss and gradient (the generator compared all four sizes bit for bit when it wrote the file)."""
    from oracle import eht_synth
    from oracle.bh_oracle import closure_tables
    g = golden("bh_sizes")
    n, tag = 64, "n64"
    syn = eht_synth.synthetic_obs(int(g["n_sites"]), 64, seed=int(g["obs_seed"]), sigma_scale=float(g[tag + "_sigma_scale"]))
    assert np.array_equal(syn["uv"], g[tag + "_uv"])
    A = np.stack([eht_synth.ftmatrix_dense(eht_synth.PSIZE, n, n, syn["uv"][i]) for i in range(64)])
    cpi, cpc, cai, cac = closure_tables(syn["times"], syn["t1"], syn["t2"], syn["tri"], syn["quad"])
    assert np.array_equal(np.asarray(cpi), g[tag + "_cp_index"]) and np.array_equal(np.asarray(cai), g[tag + "_camp_index"])
    orc = BlackholeOracle(torch.from_numpy(A).to(torch.complex64).unsqueeze(0), torch.from_numpy(syn["sigma"]).float() / 0.001,
                          torch.tensor(cpi), torch.tensor(cpc), torch.tensor(cai), torch.tensor(cac), 64, n, n,
                          **weights_kw(g))
    x = torch.from_numpy(eht_synth.ring_video(400 + n, n=n))
    assert float(x.double().sum()) == float(g[tag + "_x_checksum"])
    y = obs_from_golden(g, tag + "_y_")
    for name, got, want in zip(Obs._fields, orc(x), obs_from_golden(g, tag + "_obs_")):
        close(got, want, 1e-5)
    grad, lsum = orc.gradient(x, y, return_loss=True)
    close(lsum, g[tag + "_loss_sum"], 1e-5)
    close(grad.flatten()[::97], g[tag + "_grad_sub"], 1e-4)
