"""CPU: host logic of the sampler layer (registries, learning-rate decay, schedules, recorder,
kernel parameter packing) -- no kernel launches."""
import warnings

import numpy as np
import pytest
import torch

from step_amd.forward_operator import base as op_base
from step_amd.sampler import LatentDAPS, MCMCSampler, Trajectory, get_sampler, register_sampler
from step_amd.sampler.schedule import EDMScheduler, get_diffusion_scheduler, pfode_euler
from tests.helpers import t
from tests.toy_model import ToyModel


def test_registries_behave_like_the_reference():
    with pytest.raises(NameError, match="not defined"):
        op_base.get_operator("nope")
    with pytest.raises(NameError, match="not defined"):
        get_sampler("nope")
    assert op_base.__OPERATOR__["blackhole"].name == "blackhole" and op_base.__OPERATOR__["mri"].name == "mri"

    @register_sampler("tmp_sampler")
    class A:
        pass
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")

        @register_sampler("tmp_sampler")
        class B:
            pass
    assert any("already registered" in str(x.message) for x in w)
    assert isinstance(get_sampler("tmp_sampler"), B)


def test_get_lr_and_params(golden):
    m = golden("mcmc_bh_small")
    s = MCMCSampler(num_steps=6, lr=1.2e-5, tau=0.021, lr_min_ratio=0.015, mc_algo="hmc", momentum=0.0)
    assert s.get_lr(float(m["ratio"])) == float(m["hmc_exact_latent_lr"])
    assert s.get_lr(0.0) == 1.2e-5 and abs(s.get_lr(1.0) - 1.2e-5 * 0.015) < 1e-20
    p = s._params(s.get_lr(0.25), torch.tensor(0.7))
    lr = s.get_lr(0.25)
    assert p.algo == 1 and p.sqrt_lr == np.float32(np.sqrt(lr)) and p.noise_scale == np.float32(np.sqrt(2.0))
    assert p.inv_tau2 == np.float32(1 / 0.021 ** 2) and abs(p.inv_sigma2 - 1 / 0.49) < 1e-6
    with pytest.raises(NotImplementedError):
        MCMCSampler(num_steps=1, lr=1e-3, mc_algo="nuts")._params(1e-3, 1.0)
    with pytest.raises(RuntimeError, match="GPU only"):
        MCMCSampler._kernel_tensor(torch.zeros(2), torch.zeros(2))


def test_edm_schedule_matches_reference(golden):
    e = golden("edm_sigmas")
    assert torch.equal(EDMScheduler(25).sigma_steps, t(e["s25"]))
    assert torch.equal(get_diffusion_scheduler(name="edm", num_steps=20).sigma_steps, t(e["s20"]))
    with pytest.raises(NameError):
        get_diffusion_scheduler(name="vp", num_steps=3)
    # Euler PF-ODE on the analytic Gaussian prior of the toy model: x(t) = x0 sqrt((1+t^2)/(1+T^2))
    model = ToyModel((2, 1, 4, 4))
    sched = EDMScheduler(200, sigma_max=3.0, sigma_min=1e-3)
    x = torch.randn(2, 2, 1, 4, 4)
    out = pfode_euler(model, sched, x)
    assert torch.allclose(out, x * (1 / 10) ** 0.5, rtol=2e-2, atol=1e-3)


def test_trajectory_recorder_and_daps_construction():
    tr = Trajectory()
    for i in range(3):
        tr.add_tensor("xi", torch.full((2, 4), float(i)))
        tr.add_value("loss", float(i))
    tr.compile()
    assert tr.tensor_data["xi"].shape == (3, 2, 4) and tr.value_data["loss"].tolist() == [0.0, 1.0, 2.0]
    m = Trajectory.merge([tr, tr])
    assert m.tensor_data["xi"].shape == (3, 4, 4)
    cfg = dict(annealing_scheduler_config=dict(name="edm", num_steps=25, sigma_max=100, sigma_min=1e-2, timestep="poly-7"),
               diffusion_scheduler_config=dict(name="edm", num_steps=20, sigma_min=1e-5, timestep="poly-7", sigma_max=5),
               mcmc_sampler_config=dict(prior_solver="exact", mc_algo="hmc", num_steps=60, lr=1.2e-5, tau=0.021,
                                        lr_min_ratio=0.015, momentum=0.0), batch_consistency=False)
    s = get_sampler("latent_daps", **cfg)
    assert isinstance(s, LatentDAPS) and "sigma_max" not in s.diffusion_scheduler_config
    assert s.annealing_scheduler.num_steps == 26 and s.mcmc_sampler.num_steps == 60
    ref = torch.zeros(3, 5, 1, 2, 2)
    torch.manual_seed(0)
    assert s.get_start(ref).std() > 50
    s.batch_consistency = True
    n = s.get_randn_like(ref)
    assert torch.equal(n[:, 0], n[:, 4])
