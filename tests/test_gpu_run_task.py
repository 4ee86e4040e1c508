"""The task runner (step_amd/run_task.py, the role of the reference's sample.py:18-86) end to end
on the GPU at the tasks' own shapes with shortened schedules: operator by name from the task
configuration, measure(), LatentDAPS through LatentWrapper with the stand-in prior, chains batched,
the reference's on-disk layout."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("task,shape", [("blackhole", (64, 1, 256, 256)), ("mri", (12, 2, 192, 192))])
def test_run_task_end_to_end(task, shape, tmp_path, dev):
    from step_amd import run_task
    from step_amd.forward_operator.blackhole import load_observation
    out = tmp_path / task
    stats = run_task.main(["--task", task, "--num-samples", "3", "--num-videos", "2", "--out", str(out),
                           "--annealing-steps", "3", "--mcmc-steps", "4", "--diffusion-steps", "2",
                           "--max-batch", "2"])
    assert len(stats) == 2 and all(s["finite"] and s["samples"] == 3 for s in stats)
    # every sample went through the task's evaluator (sample.py:71)
    want = "chisq" if task == "blackhole" else "psnr"
    for s in stats:
        assert s["metric"] == want and s["metric_min"] <= s["metric_mean"] <= s["metric_max"]
        assert all(v == v for v in (s["metric_min"], s["metric_max"], s["metric_std"]))
    for idx in (0, 1):
        d = out / ("%06d" % idx)
        assert sorted(os.listdir(d)) == ["observations.pth", "sample_0.pth", "sample_1.pth", "sample_2.pth", "videos.pth"]
        s = torch.load(d / "sample_1.pth", weights_only=True)
        assert s.shape == (1,) + shape and torch.isfinite(s).all()
        if task == "blackhole":
            cp = load_observation(d / "observations.pth").cphase
            assert cp.shape[0] == 1 and cp.shape[1] % 64 == 0 and cp.shape[1] > 0
    a = torch.load(out / "000000" / "sample_0.pth", weights_only=True)
    b = torch.load(out / "000000" / "sample_1.pth", weights_only=True)
    c = torch.load(out / "000001" / "sample_0.pth", weights_only=True)
    assert not torch.equal(a, b) and not torch.equal(a, c)


def test_reference_style_yaml_is_accepted(tmp_path):
    from step_amd import run_task
    p = tmp_path / "task.yaml"
    p.write_text("""
test_idx: [0, 1]
diffusion_wrapper:
  name: st_diffusion
  video_shape: [12, 2, 192, 192]
  latent_shape: [12, 2, 48, 48]
forward_operator:
  name: mri
  acc_factor: 8
  mode: split_mask
  resolution: 192
  frames: 12
  num_acs_lines: 12
sampler:
  name: 'latent_daps'
  annealing_scheduler_config: {name: 'edm', num_steps: 5, sigma_max: 100, sigma_min: 1e-2, timestep: 'poly-7'}
  diffusion_scheduler_config: {name: 'edm', num_steps: 4, sigma_min: 1e-5, timestep: 'poly-7'}
  mcmc_sampler_config: {prior_solver: 'gaussian', mc_algo: 'langevin', num_steps: 3, lr: 1.2e-3, tau: 0.10, lr_min_ratio: 0.011, momentum: 0.83}
  batch_consistency: False
""")
    cfg = run_task.load_task("mri", str(p))
    assert cfg["sampler"]["annealing_scheduler_config"]["sigma_min"] == 1e-2
    op = run_task.build_operator(cfg, "cuda:0")
    assert op.acc_factor == 8 and op.mode == "split_mask"
    stats = run_task.main(["--task", "mri", "--config", str(p), "--num-samples", "2"])
    assert stats[0]["finite"]


@pytest.mark.parametrize("extra", [["--pixel-space"], ["--noise", "torch"], ["--prior-solver", "gaussian", "--max-batch", "2"]])
@pytest.mark.parametrize("task", ["blackhole", "mri"])
def test_run_task_variants(task, extra, dev):
    """The runner's other routes end to end: optimize() on decoded frames (the operators' in-kernel
    descent where there is one), torch noise, the 'gaussian' prior solver with chunked chains."""
    from step_amd import run_task
    stats = run_task.main(["--task", task, "--num-samples", "3", "--annealing-steps", "2", "--mcmc-steps", "3",
                           "--diffusion-steps", "2"] + extra)
    assert len(stats) == 1 and stats[0]["finite"] and stats[0]["samples"] == 3
    assert stats[0]["metric_min"] == stats[0]["metric_min"]                 # not NaN


@pytest.mark.parametrize("task,extra", [("blackhole", []), ("mri", []), ("mri", ["--pixel-space", "--prior-solver", "gaussian"]),
                                        ("blackhole", ["--pixel-space", "--prior-solver", "gaussian", "--noise", "torch"])])
def test_run_task_at_the_per_gpu_batches_of_configs_4_and_5(task, extra, dev):
    """BASELINE configs 4 / 5 on one of 8 GPUs: 64 / 8 = 8 black-hole chains and 32 / 8 = 4 MRI chains in
    ONE batch (the batches at which the two-launch black-hole step and the shared-frame MRI kernel run),
    at the shipped MCMC settings with a shortened annealing schedule; and the pixel-space routes, whose
    steps are the operators' in-kernel updates (descent_step / langevin_step)."""
    from step_amd import run_task
    n = 8 if task == "blackhole" else 4
    stats = run_task.main(["--task", task, "--num-samples", str(n), "--max-batch", str(n), "--annealing-steps", "3",
                           "--mcmc-steps", "12", "--diffusion-steps", "2"] + extra)
    assert len(stats) == 1 and stats[0]["finite"] and stats[0]["samples"] == n
    assert stats[0]["metric_min"] == stats[0]["metric_min"]


def test_run_task_at_another_mri_resolution(tmp_path, dev):
    """The reference's operator takes any `resolution` (dynamic_mri.py:9): a task file at 128 x 128 runs end to end
    on the any-size kernels (csrc/mri_any.hip), latent and pixel-space routes."""
    from step_amd import run_task
    p = tmp_path / "task128.yaml"
    p.write_text("""
test_idx: [0]
diffusion_wrapper:
  name: st_diffusion
  video_shape: [6, 2, 128, 128]
  latent_shape: [6, 2, 32, 32]
forward_operator:
  name: mri
  acc_factor: 4
  mode: random_split_mask
  resolution: 128
  frames: 6
  num_acs_lines: 16
sampler:
  name: 'latent_daps'
  annealing_scheduler_config: {name: 'edm', num_steps: 3, sigma_max: 100, sigma_min: 1e-2, timestep: 'poly-7'}
  diffusion_scheduler_config: {name: 'edm', num_steps: 2, sigma_min: 1e-5, timestep: 'poly-7'}
  mcmc_sampler_config: {prior_solver: 'gaussian', mc_algo: 'langevin', num_steps: 4, lr: 1.2e-3, tau: 0.10, lr_min_ratio: 0.011}
""")
    for extra in ([], ["--pixel-space"]):
        stats = run_task.main(["--task", "mri", "--config", str(p), "--num-samples", "3"] + extra)
        assert len(stats) == 1 and stats[0]["finite"] and stats[0]["samples"] == 3
        assert stats[0]["metric"] == "psnr" and stats[0]["metric_min"] == stats[0]["metric_min"]


def test_run_task_sharded_over_two_ranks_on_one_card(tmp_path, dev):
    """`torchrun --nproc-per-node 2 -m step_amd.run_task ... --rehearse-on-one-gpu`: the sharded path of configs
    4 / 5 (chains dealt to ranks, per-chain seeds, gather to rank 0, rank 0 writes) with both ranks on cuda:0 and the
    process group on gloo; the samples must equal a one-process run's (a chain's draws do not depend on the world
    size)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--task", "mri", "--num-samples", "5", "--annealing-steps", "3", "--mcmc-steps", "4", "--diffusion-steps", "2"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29549", "-m", "step_amd.run_task"] + common +
                       ["--out", str(tmp_path / "two"), "--rehearse-on-one-gpu"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=400, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    from step_amd import run_task
    run_task.main(common + ["--out", str(tmp_path / "one")])
    for i in range(5):
        a = torch.load(tmp_path / "two" / "000000" / ("sample_%d.pth" % i), weights_only=True)
        b = torch.load(tmp_path / "one" / "000000" / ("sample_%d.pth" % i), weights_only=True)
        assert not a.requires_grad and not b.requires_grad
        # the same chain from the same seeds; the stand-in decoder (stock convolutions) sees 3 + 2 chains per batch
        # instead of 5 and may pick other algorithms, so the comparison is to rounding, not to the bit
        err = ((a - b).abs().max() / b.abs().max()).item()
        assert err <= 1e-3, "sample %d differs between 2 ranks and 1: %.2e" % (i, err)
