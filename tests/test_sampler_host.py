"""CPU: host logic of the sampler layer (registries, learning-rate decay, schedules, recorder,
kernel parameter packing) -- no kernel launches."""
import os
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


def test_observation_pickle_roundtrip_with_reference_identity(tmp_path):
    """observations.pth compatibility (reference sample.py:62): files carry the reference's class
    identity and load through torch's weights-only unpickler; a file written by a class that only
    looks like the reference's (same module path, same attribute names) loads too."""
    import pickle
    import sys
    import types
    import torch
    from step_amd.forward_operator.blackhole import (BlackHoleObservation, load_observation,
                                                     save_observation)
    g = torch.Generator().manual_seed(0)
    fields = [torch.randn(2, 5, generator=g) for _ in range(9)]
    fields[0] = torch.complex(fields[0], fields[1])
    obs = BlackHoleObservation(*fields)
    path = tmp_path / "observations.pth"
    save_observation(obs, path)
    assert "forward_operator.blackhole" not in sys.modules          # the temporary stub is gone
    raw = open(path, "rb").read()
    assert b"forward_operator.blackhole" in raw and b"step_amd" not in raw
    back = load_observation(path)
    assert type(back) is BlackHoleObservation
    for a, b in zip(obs.to_list(), back.to_list()):
        assert torch.equal(a, b)

    # a stand-in for the reference's module, written with plain torch.save as sample.py does
    mod = types.ModuleType("forward_operator.blackhole")

    class BlackHoleObservation2(object):
        def __init__(self, *f):
            (self.vis, self.sigma_vis, self.amp, self.sigma_amp, self.cphase, self.sigma_cphase,
             self.logcamp, self.sigma_logcamp, self.flux) = f
    BlackHoleObservation2.__module__ = "forward_operator.blackhole"
    BlackHoleObservation2.__qualname__ = BlackHoleObservation2.__name__ = "BlackHoleObservation"
    mod.BlackHoleObservation = BlackHoleObservation2
    sys.modules["forward_operator"] = types.ModuleType("forward_operator")
    sys.modules["forward_operator.blackhole"] = mod
    try:
        torch.save(BlackHoleObservation2(*fields), tmp_path / "ref.pth")
        save_observation(obs, tmp_path / "ours_inside_ref.pth")     # uses the loaded module's class
    finally:
        del sys.modules["forward_operator.blackhole"], sys.modules["forward_operator"]
    for name in ("ref.pth", "ours_inside_ref.pth"):
        back = load_observation(tmp_path / name)
        for a, b in zip(fields, back.to_list()):
            assert torch.equal(a, b)
    # anything else in the file is refused by the weights-only loader
    with open(tmp_path / "evil.pth", "wb") as fh:
        pickle.dump(types.SimpleNamespace(a=1), fh)
    with pytest.raises(Exception):
        load_observation(tmp_path / "evil.pth")


def test_write_sample_dir_layout(tmp_path):
    import torch
    from step_amd.driver import write_sample_dir
    from step_amd.forward_operator.blackhole import BlackHoleObservation, load_observation
    obs = BlackHoleObservation(*[torch.full((1, 3), float(i)) for i in range(9)])
    samples = torch.arange(24.0).view(3, 2, 1, 2, 2)
    write_sample_dir(tmp_path / "0007", obs, samples, videos=samples[:1])
    names = sorted(p.name for p in (tmp_path / "0007").iterdir())
    assert names == ["observations.pth", "sample_0.pth", "sample_1.pth", "sample_2.pth", "videos.pth"]
    assert torch.equal(torch.load(tmp_path / "0007" / "sample_2.pth", weights_only=True), samples[2:3])
    assert torch.equal(load_observation(tmp_path / "0007" / "observations.pth").flux, obs.flux)


def test_blackhole_observation_container_semantics():
    """a18: the nine-field container behaves like reference forward_operator/blackhole.py:23-132."""
    import torch
    from step_amd.forward_operator import BlackHoleObservation
    names = ("vis", "sigma_vis", "amp", "sigma_amp", "cphase", "sigma_cphase", "logcamp",
             "sigma_logcamp", "flux")
    f = [torch.full((3, 4), float(i)) for i in range(9)]
    o = BlackHoleObservation(*f)
    assert len(o) == 3
    assert [float(x[0, 0]) for x in o.to_list()] == list(range(9))          # to_list order
    assert list(o.to_dict()) == list(names)
    assert o[1:3].amp.shape == (2, 4) and o[0].flux.shape == (4,)
    m = BlackHoleObservation.merge([o, o[0:1]])
    assert len(m) == 4 and torch.equal(m.cphase[3], o.cphase[0])
    assert o.to("cpu").vis is not None and type(o.to("cpu")) is BlackHoleObservation
    g = torch.ones(2, 2, requires_grad=True)
    p = BlackHoleObservation(cphase=g * 2, sigma_cphase=g * 3)              # partial observation
    assert set(p.to_dict()) == {"cphase", "sigma_cphase"} and p[0].vis is None
    assert p.detach() is None and not p.cphase.requires_grad                # in place, like the reference
    q = BlackHoleObservation().from_dict({"flux": f[8], "amp": f[2], "sigma_amp": f[3]})
    assert q.flux is f[8] and q.amp is f[2] and q.vis is None


def test_scheduler_takes_the_reference_yaml_as_pyyaml_reads_it():
    """PyYAML reads the reference configs' `sigma_min: 1e-2` / `1e-5` as STRINGS; a caller that
    splats yaml.safe_load(...)['sampler'][...] into get_diffusion_scheduler (sample.py:52) must get
    the same grid as with floats, and tensor sigmas must stay tensors (float32 arithmetic)."""
    import yaml
    from step_amd.sampler.schedule import get_diffusion_scheduler
    cfg = yaml.safe_load("annealing_scheduler_config:\n  name: edm\n  num_steps: 25\n  sigma_max: 100\n"
                         "  sigma_min: 1e-2\n  timestep: poly-7\n")["annealing_scheduler_config"]
    assert isinstance(cfg["sigma_min"], str) and isinstance(cfg["sigma_max"], int)
    a = get_diffusion_scheduler(**cfg)
    b = get_diffusion_scheduler(name="edm", num_steps=25, sigma_max=100.0, sigma_min=1e-2, timestep="poly-7")
    assert torch.equal(a.sigma_steps, b.sigma_steps) and a.sigma_steps[-1] == 0
    c = get_diffusion_scheduler(name="edm", num_steps=3, sigma_max=torch.tensor(2.0), sigma_min="1e-5")
    assert torch.is_tensor(c.sigma_max) and c.sigma_steps.dtype == torch.float32


def test_run_task_array_file_resolution(tmp_path):
    import warnings
    from step_amd.run_task import resolve_array
    (tmp_path / "configs" / "task").mkdir(parents=True)
    cfgf = tmp_path / "configs" / "task" / "blackhole.yaml"
    cfgf.write_text("x: 1\n")
    arr = tmp_path / "configs" / "task" / "EHT2017.txt"
    arr.write_text("#\n")
    # the reference's relative path, started from another directory: found next to the config
    got, standin = resolve_array("configs/task/EHT2017.txt", str(cfgf))
    assert os.path.samefile(got, arr) and not standin
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got, standin = resolve_array("configs/task/NOPE.txt", str(cfgf))
    assert standin and got.endswith("array_8.txt") and any("stand-in" in str(m.message) for m in w)
