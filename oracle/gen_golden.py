#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own code (build container only).

Run:  python oracle/gen_golden.py           (needs /root/reference; never runs on the GPU box)

The reference's Python modules are imported from /root/reference with stub modules
standing in for the third-party packages that are absent here (ehtim, piq,
torchdiffeq); only stubs for *absent third-party names* are inserted -- every
arithmetic line that produces a fixture value is the reference's own.  The ehtim
entry points the reference calls at set-up (eh.array.load_txt -> .obsdata(),
obsh.ftmatrix, Obsdata.tlist/c_phases/c_amplitudes) are served by a synthetic
observation table (oracle/eht_synth.py, PARITY UNPINNED) so that the reference's
own ``process_obs`` loop (blackhole.py:183-300) builds the index tensors.

Fixtures are plain arrays: inputs, expected outputs.  While generating, every
value is also compared with the oracle restatement (oracle/*.py) and the script
aborts on a mismatch, which is what pins the oracle.
"""
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)

from oracle import eht_synth                      # noqa: E402
from oracle.bh_oracle import BlackholeOracle, Obs, closure_tables  # noqa: E402
from oracle.mri_oracle import MRIOracle           # noqa: E402
from oracle.mcmc_oracle import MCMCOracle, LatentWrapOracle, edm_sigmas, lr_at  # noqa: E402
from tests.toy_model import ToyModel              # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")


# --------------------------------------------------------------------------- stubs
def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _FakeObsdata:
    """Stands in for ehtim.obsdata.Obsdata: holds a synthetic table."""

    def tlist(self, conj=False, t_gather=0., scan_gather=False):
        d = self.data
        return [d[d["time"] == t] for t in np.unique(d["time"])]

    def c_phases(self):
        return self._tri

    def c_amplitudes(self):
        return self._quad


class _FakeArray:
    def __init__(self, syn):
        self.syn = syn

    def obsdata(self, *a, **k):
        s = self.syn
        n = len(s["times"])
        data = np.zeros(n, dtype=[("time", "f8"), ("t1", "U8"), ("t2", "U8"),
                                  ("u", "f8"), ("v", "f8"), ("sigma", "f8")])
        data["time"], data["t1"], data["t2"] = s["times"], s["t1"], s["t2"]
        uv = s["uv"].reshape(-1, 2)
        data["u"], data["v"], data["sigma"] = uv[:, 0], uv[:, 1], s["sigma"]
        tri = np.array(s["tri"], dtype=[("time", "f8"), ("t1", "U8"), ("t2", "U8"), ("t3", "U8")])
        quad = np.array(s["quad"], dtype=[("time", "f8"), ("t1", "U8"), ("t2", "U8"),
                                          ("t3", "U8"), ("t4", "U8")])
        o = _FakeObsdata()
        o.data, o._tri, o._quad = data, tri, quad
        return o


_CUR = {}


def install_stubs():
    eh = _stub("ehtim", DEGREE=np.pi / 180.0)
    eh.array = types.SimpleNamespace(load_txt=lambda path: _FakeArray(_CUR["syn"]))
    _stub("ehtim.statistics")
    _stub("ehtim.statistics.dataframes")
    _stub("ehtim.observing")
    _stub("ehtim.observing.obs_helpers",
          ftmatrix=lambda psize, xd, yd, uv, pulse=None, mask=[]:
          eht_synth.ftmatrix_dense(psize, xd, yd, uv))
    _stub("ehtim.obsdata", Obsdata=_FakeObsdata)
    _stub("ehtim.observing.pulses", trianglePulse2D="trianglePulse2D")
    _stub("piq", psnr=None, ssim=None, LPIPS=None)
    _stub("torchdiffeq", odeint=None, odeint_adjoint=None)
    # the reference hard-codes .cuda() on the index tensors (blackhole.py:251-264)
    torch.Tensor.cuda = lambda self, *a, **k: self


def import_reference():
    os.chdir(REF)
    sys.path.insert(0, REF)
    import forward_operator as fo                      # noqa
    from forward_operator.latent_wrapper import LatentWrapper
    from sampler.daps import MCMCSampler
    from cores.scheduler import get_diffusion_scheduler
    return fo, LatentWrapper, MCMCSampler, get_diffusion_scheduler


def close(a, b, tol=0.0, what=""):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    if tol == 0.0:
        ok = torch.equal(a, b) or (torch.isnan(a) == torch.isnan(b)).all() and \
            torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))
    else:
        ok = torch.allclose(a, b, rtol=tol, atol=0)
    if not ok:
        raise SystemExit("oracle != reference for %s (max abs diff %g)" %
                         (what, (a - b).abs().max().item()))


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %-28s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


# --------------------------------------------------------------------------- blackhole
def ref_blackhole(fo, syn, imsize, **kw):
    _CUR["syn"] = syn
    return fo.get_operator("blackhole", array="synthetic", imsize=imsize, num_frames=64,
                           device="cpu", **kw)


def oracle_from_ref(op, kw):
    return BlackholeOracle(op.A_vis, op.sigma, op.cp_index, op.cp_conjugate, op.camp_index,
                           op.camp_conjugate, op.T, op.H, op.W, **kw)


def obs_arrays(prefix, o):
    fields = o.to_list() if hasattr(o, "to_list") else tuple(o)
    return {prefix + n: f for n, f in zip(Obs._fields, fields)}


def gen_bh_case(fo, LatentWrapper, name, n_sites, imsize, B, seed, weights, full=False,
                sigma_scale=1.0, toy_gain=1.5):
    # sigma_scale puts sigma/|vis| near 0.05 so that the random-x case sits on the
    # chi^2 >= 1 branch of normalize_chisq and the near-truth case on the 1/chi^2 branch
    syn = eht_synth.synthetic_obs(n_sites, 64, seed=seed, sigma_scale=sigma_scale)
    kw = dict(ref_multiplier=0.001, **weights)
    op = ref_blackhole(fo, syn, imsize, **kw)
    okw = {k: v for k, v in kw.items() if k != "ref_multiplier"}
    orc = oracle_from_ref(op, okw)

    # a14: the reference's own loop vs the restated mapping -- bit exact
    cpi, cpc, cai, cac = closure_tables(syn["times"], syn["t1"], syn["t2"], syn["tri"], syn["quad"])
    close(op.cp_index, torch.tensor(cpi), what="cp_index")
    close(op.cp_conjugate, torch.tensor(cpc), what="cp_conjugate")
    close(op.camp_index, torch.tensor(cai), what="camp_index")
    close(op.camp_conjugate, torch.tensor(cac), what="camp_conjugate")

    g = torch.Generator().manual_seed(seed)
    x_gt = torch.rand(B, 64, 1, imsize, imsize, generator=g) * 2 - 1
    x = torch.rand(B, 64, 1, imsize, imsize, generator=g) * 2 - 1
    x_near = (x_gt + 2e-3 * torch.randn(x_gt.shape, generator=g)).clamp(-1, 1)
    y = op(x_gt)                                   # BlackHoleObservation
    yo = Obs(*y.to_list())

    out = dict(n_sites=n_sites, imsize=imsize, seed=seed, sigma_scale=sigma_scale,
               toy_gain=toy_gain, uv=syn["uv"], sigma_raw=syn["sigma"],
               sigma=op.sigma, cp_index=op.cp_index, cp_conjugate=op.cp_conjugate,
               camp_index=op.camp_index, camp_conjugate=op.camp_conjugate,
               weights=np.array([weights.get(k, d) for k, d in
                                 (("w_vis", 0), ("w_amp", 0), ("w_cphase", 1), ("w_logcamp", 1),
                                  ("w_flux", 0.5))], dtype=np.float64),
               loss_normalize=int(weights.get("loss_normalize", True)),
               tri_sites=np.array([t[1:] for t in syn["tri"]]),
               quad_sites=np.array([q[1:] for q in syn["quad"]]),
               row_t1=np.array(syn["t1"]), row_t2=np.array(syn["t2"]), row_time=syn["times"])
    if not full:
        out.update(A_vis=op.A_vis[0], x_gt=x_gt, x=x, x_near=x_near)
    out.update(obs_arrays("y_", y))

    # fp64 evaluation of the same formulas: the yardstick where fp32 itself is noise-limited
    orc64 = BlackholeOracle(op.A_vis.to(torch.complex128), op.sigma.double(), op.cp_index,
                            op.cp_conjugate, op.camp_index, op.camp_conjugate, op.T, op.H, op.W,
                            **okw)
    y64 = Obs(*[f.to(torch.complex128) if torch.is_complex(f) else f.double() for f in y.to_list()])
    for tag, xin in (("x", x), ("near", x_near)):
        o_ref = op(xin)
        o_orc = orc(xin)
        out.update(obs_arrays(tag + "_obs64_", orc64(xin.double())))
        g64, l64 = orc64.gradient(xin.double(), y64, return_loss=True)
        out.update({tag + "_loss64": orc64.loss(xin.double(), y64)})
        out.update({tag + "_grad64_sub": g64.flatten()[::97]} if full else {tag + "_grad64": g64})
        for n, a, b in zip(Obs._fields, o_ref.to_list(), o_orc):
            close(a, b, what="%s obs.%s" % (name, n))
        loss = op.loss(xin, y)
        close(loss, orc.loss(xin, yo), what=name + " loss")
        grad, lsum = op.gradient(xin, y, return_loss=True)
        g2, l2 = orc.gradient(xin, yo, return_loss=True)
        close(grad, g2, what=name + " gradient")
        close(lsum, l2, what=name + " loss sum")
        x01 = op.unnormalize(xin)
        chis = dict(chi2_vis=op.chi2_vis(x01, y.vis, y.sigma_vis),
                    chi2_amp=op.chi2_amp(x01, y.amp, y.sigma_amp),
                    chi2_cphase=op.chi2_cphase(x01, y.cphase, y.sigma_cphase),
                    chi2_logcamp=op.chi2_logcamp(x01, y.logcamp, y.sigma_logcamp),
                    chi2_flux=op.chi2_flux(x01, y.flux))
        ev = op.evaluate_chisq(xin, y, normalize=False)
        evn = op.evaluate_chisq(xin, y, normalize=True, chi_sq_list=["cphase", "logcamp"])
        ev_o = orc.evaluate_chisq(xin, yo, normalize=False)
        for k in ev:
            close(ev[k], ev_o[k], what=name + " evaluate_chisq " + k)
        if full:
            out.update({tag + "_grad_sub": grad.flatten()[::97], tag + "_grad_sum": grad.sum(),
                        tag + "_grad_abs_sum": grad.abs().sum(),
                        tag + "_grad_frame_sum": grad.flatten(2).sum(-1)})
        else:
            out.update({tag + "_grad": grad})
        out.update(obs_arrays(tag + "_obs_", o_ref))
        out.update({tag + "_loss": loss, tag + "_loss_sum": lsum})
        out.update({tag + "_" + k: v for k, v in chis.items()})
        out.update({tag + "_eval_" + k: v for k, v in ev.items()})
        out.update({tag + "_evaln_" + k: v for k, v in evn.items()})

    if not full:
        # a7: latent wrapper with the toy decoder (incl. the +-1 clamp)
        model = ToyModel((64, 1, imsize // 2, imsize // 2), factor=2, gain=toy_gain)
        z = 0.5 * torch.randn(B, 64, 1, imsize // 2, imsize // 2, generator=g)
        lw = LatentWrapper(op, model)
        zg, zl = lw.gradient(z, y, return_loss=True)
        zg2, zl2 = LatentWrapOracle(orc, model).gradient(z, yo, return_loss=True)
        close(zg, zg2, what=name + " latent gradient")
        close(zl, zl2, what=name + " latent loss")
        out.update(z=z, z_grad=zg, z_loss_sum=zl, z_clamped=int((zg.abs() == 1).sum()))
    npz(name, **out)
    return op, orc, y, yo, syn


# --------------------------------------------------------------------------- MRI
def gen_mri(fo):
    masks = {}
    cfgs = [(6, 24, 192, 12), (8, 12, 192, 12), (4, 8, 64, 5), (6, 24, 192, 7)]
    for acc, acs, res, fr in cfgs:
        for mode in ("same_mask", "split_mask", "random_split_mask"):
            op = fo.get_operator("mri", acc_factor=acc, mode=mode, resolution=res, frames=fr,
                                 num_acs_lines=acs)
            orc = MRIOracle(acc, mode, res, fr, acs)
            close(op.mask, orc.mask, what="mri mask %s" % mode)
            masks["%s_a%d_c%d_r%d_f%d" % (mode, acc, acs, res, fr)] = \
                op.mask[0, :, 0, :, 0].to(torch.uint8)
    npz("mri_masks", **masks)

    for mode, B, seed, noise in (("same_mask", 2, 37, 0.0), ("split_mask", 1, 38, 0.05),
                                 ("random_split_mask", 1, 39, 0.05)):
        op = fo.get_operator("mri", acc_factor=6, mode=mode, resolution=192, frames=12,
                             num_acs_lines=24, sigma=noise)
        orc = MRIOracle(6, mode, 192, 12, 24, sigma=noise)
        rng = np.random.RandomState(seed)
        x_gt = torch.from_numpy(rng.randn(B, 12, 2, 192, 192).astype(np.float32))
        x = torch.from_numpy(rng.randn(B, 12, 2, 192, 192).astype(np.float32))
        nz = torch.from_numpy(rng.randn(B, 12, 2, 192, 192).astype(np.float32))
        y = op(x_gt) + noise * nz                      # base.py:36-38 with explicit noise
        fwd = op(x)
        close(fwd, orc(x), what="mri forward")
        grad, lsum = op.gradient(x, y, return_loss=True)
        g2, l2 = orc.gradient(x, y, return_loss=True)
        close(grad, g2, what="mri gradient")
        close(lsum, l2, what="mri loss")
        loss = op.loss(x, y)
        npz("mri_%s" % mode, seed=seed, B=B, noise=noise,
            mask=op.mask[0, :, 0, :, 0].to(torch.uint8),
            fwd_sub=fwd.flatten()[::11], fwd_sq_sum=(fwd.double() ** 2).flatten(1).sum(-1),
            fwd_frame0=fwd[0, 0],                       # one full frame [2,192,192]
            grad_sub=grad.flatten()[::11], grad_sq_sum=(grad.double() ** 2).flatten(1).sum(-1),
            grad_frame0=grad[0, 0], loss=loss, loss_sum=lsum)


# --------------------------------------------------------------------------- MCMC
class NoiseTape:
    """Records every torch.randn_like / rand_like draw (order = reference's order)."""

    def __init__(self):
        self.items = []
        self._rn, self._ru = torch.randn_like, torch.rand_like

    def __enter__(self):
        def rn(t, *a, **k):
            v = self._rn(t, *a, **k)
            self.items.append(v.clone())
            return v

        def ru(t, *a, **k):
            v = self._ru(t, *a, **k)
            self.items.append(v.clone())
            return v
        torch.randn_like, torch.rand_like = rn, ru
        return self

    def __exit__(self, *a):
        torch.randn_like, torch.rand_like = self._rn, self._ru


def gen_mcmc(MCMCSampler, LatentWrapper, op, orc, y, yo, imsize):
    torch.manual_seed(37)
    B, h = 2, imsize // 2
    model = ToyModel((64, 1, h, h), factor=2, gain=1.5)
    z0hat = 0.3 * torch.randn(B, 64, 1, h, h)
    zt = z0hat + 0.5 * torch.randn(B, 64, 1, h, h)
    x0hat = model.decode(z0hat).detach()
    out = dict(z0hat=z0hat, zt=zt, x0hat=x0hat)

    cases = {
        # shipped blackhole-style settings (hmc, exact prior) through the latent wrapper
        "hmc_exact_latent": (dict(num_steps=6, lr=1.2e-5, tau=0.021, lr_min_ratio=0.015,
                                  prior_solver="exact", mc_algo="hmc", momentum=0.0), "latent"),
        "hmc_mom_gauss_latent": (dict(num_steps=6, lr=1.2e-3, tau=0.5, lr_min_ratio=0.011,
                                      prior_solver="gaussian", mc_algo="hmc", momentum=0.83),
                                 "latent"),
        "langevin_scoret_latent": (dict(num_steps=5, lr=1e-4, tau=0.3, prior_solver="score-t",
                                        mc_algo="langevin"), "latent"),
        "langevin_scoremin_pixel": (dict(num_steps=4, lr=1e-6, tau=1.0, prior_solver="score-min",
                                         mc_algo="langevin"), "pixel"),
        "mh_pixel": (dict(num_steps=6, lr=1e-7, tau=3.0, mc_algo="MH"), "pixel"),
    }
    sigma, ratio = 0.7, 0.25
    for cname, (cfg, space) in cases.items():
        ref_s = MCMCSampler(**cfg)
        orc_s = MCMCOracle(**cfg)
        if space == "latent":
            r_op, o_op = LatentWrapper(op, model), LatentWrapOracle(orc, model)
            start, xt = z0hat, zt
        else:
            r_op, o_op = op, orc
            start, xt = x0hat, x0hat + 0.1 * torch.randn_like(x0hat)
            out[cname + "_xt"] = xt
        st = torch.get_rng_state()
        with NoiseTape() as tape:
            res = ref_s.sample(xt, model, start, r_op, y, sigma, ratio, record=(cfg["mc_algo"] != "MH"))
        torch.set_rng_state(st)
        res_o = orc_s.sample(xt, model, start, o_op, yo, sigma, ratio)
        close(res, res_o, what="mcmc " + cname)
        out[cname + "_result"] = res
        out[cname + "_noise"] = torch.stack(tape.items) if len({tuple(t.shape) for t in tape.items}) == 1 \
            else torch.stack([t for t in tape.items if t.shape == tape.items[0].shape])
        if cfg["mc_algo"] == "MH":
            out[cname + "_unif"] = torch.stack([t for t in tape.items if t.dim() == 1])
        else:
            out[cname + "_traj"] = ref_s.trajectory.compile().tensor_data["xi"]
        out[cname + "_lr"] = ref_s.get_lr(ratio)
        assert ref_s.get_lr(ratio) == lr_at(cfg["lr"], cfg.get("lr_min_ratio", 0.01), ratio)
    # pixel-space optimize (daps.py:210-223)
    cfg = dict(num_steps=5, lr=2e-7, tau=1.0)
    res = MCMCSampler(**cfg).optimize(x0hat, op, y, sigma, ratio)
    close(res, MCMCOracle(**cfg).optimize(x0hat, orc, yo, sigma, ratio), what="optimize")
    out["optimize_result"] = res
    out["sigma"], out["ratio"] = sigma, ratio
    npz("mcmc_bh_small", **out)


def gen_sched(get_diffusion_scheduler):
    a = get_diffusion_scheduler(name="edm", num_steps=25, sigma_max=100, sigma_min=1e-2,
                                timestep="poly-7").sigma_steps
    b = get_diffusion_scheduler(name="edm", num_steps=20, sigma_max=100, sigma_min=1e-2,
                                timestep="poly-7").sigma_steps
    close(a, edm_sigmas(25), what="edm 25")
    close(b, edm_sigmas(20), what="edm 20")
    npz("edm_sigmas", s25=a, s20=b)


def main():
    os.makedirs(GOLD, exist_ok=True)
    install_stubs()
    fo, LatentWrapper, MCMCSampler, get_sched = import_reference()
    torch.set_num_threads(8)
    shipped = dict(w_vis=0, w_amp=0, w_cphase=1, w_logcamp=1, w_flux=0.26, loss_normalize=True)
    op, orc, y, yo, _ = gen_bh_case(fo, LatentWrapper, "bh_small", 4, 16, 2, 37, shipped,
                                    sigma_scale=0.5e-3, toy_gain=0.02)
    gen_mcmc(MCMCSampler, LatentWrapper, op, orc, y, yo, 16)
    gen_bh_case(fo, LatentWrapper, "bh_allterms", 5, 16, 3, 41,
                dict(w_vis=0.3, w_amp=0.7, w_cphase=1.5, w_logcamp=0.5, w_flux=0.26,
                     loss_normalize=False), sigma_scale=0.5e-3, toy_gain=0.01)
    gen_bh_case(fo, LatentWrapper, "bh_allterms_norm", 6, 24, 2, 43,
                dict(w_vis=0.3, w_amp=0.7, w_cphase=1.0, w_logcamp=1.0, w_flux=0.5,
                     loss_normalize=True), sigma_scale=0.7e-3, toy_gain=0.02)
    gen_mri(fo)
    gen_sched(get_sched)
    gen_bh_case(fo, LatentWrapper, "bh_full_n8", 8, 256, 1, 37, shipped, full=True,
                sigma_scale=7e-3)
    print("all fixtures written; oracle == reference on every compared value")


if __name__ == "__main__":
    main()
