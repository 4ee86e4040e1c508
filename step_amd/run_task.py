"""Posterior sampling for a task, the role of the reference's ``sample.py:18-86`` with the chains
batched per GPU and sharded over ranks (step_amd.driver):

    python -m step_amd.run_task --task blackhole --num-samples 64 --num-videos 1 --out exps/bh
    torchrun --nproc-per-node 8 -m step_amd.run_task --task mri --num-samples 32 --out exps/mri

The operator and sampler are built by name from a task configuration with the reference's keys
(``forward_operator:`` / ``sampler:`` / ``diffusion_wrapper:`` sections; YAML keys == constructor
kwargs, sample.py:48-54).  ``--config`` takes the reference's own ``configs/task/*.yaml`` unchanged;
without it the shipped values of those files are used (TASKS below).  What the reference needs
from outside and this build cannot have offline is replaced and SAID SO in the run's summary:
the pretrained prior by ``StandInPrior`` (random weights), the test videos by synthetic ones, and
ehtim by ``step_amd.eht_lite`` (array table) / ``measure()``'s restated noise model.  Every sample
is scored by the task's evaluator (``step_amd.evaluate``; sample.py:71) and the summary carries the
main metric's max / min / mean / std.

Per test video the on-disk layout is the reference's: ``{out}/{idx:06d}/observations.pth``,
``videos.pth``, ``sample_{i}.pth`` (sample.py:59-67)."""
import argparse
import json
import os
import time

import numpy as np
import torch

from . import driver
from .evaluate import get_evaluator
from .forward_operator import get_operator
from .prior_standin import StandInPrior
from .sampler import get_sampler

_HERE = os.path.dirname(os.path.abspath(__file__))

# the shipped task settings (reference configs/task/blackhole.yaml:11-60, mri.yaml:11-56)
TASKS = {
    "blackhole": dict(
        diffusion_wrapper=dict(video_shape=[64, 1, 256, 256], latent_shape=[64, 1, 32, 32]),
        forward_operator=dict(name="blackhole", array="configs/task/EHT2017.txt", imsize=256, w_vis=0, w_amp=0,
                              w_cphase=1, w_logcamp=1, w_flux=0.26, loss_normalize=True, num_frames=64,
                              ref_multiplier=0.001),
        sampler=dict(name="latent_daps",
                     annealing_scheduler_config=dict(name="edm", num_steps=25, sigma_max=100, sigma_min=1e-2, timestep="poly-7"),
                     diffusion_scheduler_config=dict(name="edm", num_steps=20, sigma_min=1e-5, timestep="poly-7"),
                     mcmc_sampler_config=dict(prior_solver="exact", mc_algo="hmc", num_steps=60, lr=1.2e-5, tau=0.021,
                                              lr_min_ratio=0.015, momentum=0.0),
                     batch_consistency=False)),
    "mri": dict(
        diffusion_wrapper=dict(video_shape=[12, 2, 192, 192], latent_shape=[12, 2, 48, 48]),
        forward_operator=dict(name="mri", acc_factor=6, mode="same_mask", resolution=192, frames=12, num_acs_lines=24),
        sampler=dict(name="latent_daps",
                     annealing_scheduler_config=dict(name="edm", num_steps=20, sigma_max=100, sigma_min=1e-2, timestep="poly-7"),
                     diffusion_scheduler_config=dict(name="edm", num_steps=20, sigma_min=1e-5, timestep="poly-7"),
                     mcmc_sampler_config=dict(prior_solver="exact", mc_algo="hmc", num_steps=53, lr=1.2e-3, tau=0.10,
                                              lr_min_ratio=0.011, momentum=0.83),
                     batch_consistency=False)),
}


def load_task(task, config=None):
    """The task's configuration: a YAML file with the reference's sections, or the shipped values."""
    if config is None:
        return json.loads(json.dumps(TASKS[task]))           # deep copy
    import yaml
    with open(config) as fh:
        cfg = yaml.safe_load(fh)
    for k in ("forward_operator", "sampler", "diffusion_wrapper"):
        if k not in cfg:
            raise KeyError("%s has no `%s:` section" % (config, k))
    for sect in ("annealing_scheduler_config", "diffusion_scheduler_config", "mcmc_sampler_config"):
        for k, v in cfg["sampler"][sect].items():            # yaml reads 1e-2 as a string
            if isinstance(v, str):
                try:
                    cfg["sampler"][sect][k] = float(v)
                except ValueError:
                    pass
    return cfg


def resolve_array(path, config=None):
    """The array file of the black-hole operator: the configured path as given, else relative to the
    config file's directory and to its parent's parent (the reference keeps `configs/task/EHT2017.txt`
    relative to the repository root while the YAML sits in configs/task/), else -- with a warning -- the
    8-station stand-in shipped here (the reference's EHT2017.txt is not redistributed).
    -> (path used, True if it is the stand-in)"""
    cands = [path] if path else []
    if path and config and not os.path.isabs(path):
        d = os.path.dirname(os.path.abspath(config))
        cands += [os.path.join(d, path), os.path.join(d, os.path.basename(path)),
                  os.path.join(os.path.dirname(os.path.dirname(d)), path)]
    for c in cands:
        if os.path.exists(c):
            return c, False
    fallback = os.path.join(_HERE, "data", "array_8.txt")
    import warnings
    warnings.warn("run_task: array file %r not found (tried %s); using the 8-station stand-in %s -- F and the "
                  "closure sets follow THAT table" % (path, ", ".join(cands) or "-", fallback))
    return fallback, True


def build_operator(cfg, device, config=None):
    kw = dict(cfg["forward_operator"])
    name = kw.pop("name")
    if name == "blackhole":
        kw["array"], standin = resolve_array(kw.get("array", ""), config)
        kw["device"] = device
        op = get_operator(name, **kw)
        op.array_file, op.array_is_standin = kw["array"], standin
        return op
    return get_operator(name, **kw)


def synthetic_video(task, idx, shape):
    """A test video in the data range of the task's dataset (the datasets are not available)."""
    T, C, H, W = shape
    rng = np.random.RandomState(1000 + idx)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    out = np.zeros((T, C, H, W), np.float32)
    for t in range(T):
        cx, cy = W / 2 + 3 * np.sin(0.1 * t + idx), H / 2 + 3 * np.cos(0.07 * t)
        r, th = np.hypot(xx - cx, yy - cy), np.arctan2(yy - cy, xx - cx)
        img = np.exp(-0.5 * ((r - 0.16 * W) / (0.03 * W)) ** 2) * (1 + 0.6 * np.cos(th - 0.09 * t - idx))
        img = img / img.max()
        if task == "blackhole":
            out[t, 0] = 2 * img - 1
        else:                                                # complex image: magnitude x smooth phase
            ph = 0.5 * np.sin(2 * np.pi * xx / W + 0.2 * t) + 0.1 * rng.randn()
            out[t, 0], out[t, 1] = img * np.cos(ph), img * np.sin(ph)
    return torch.from_numpy(out)


def main(argv=None):
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--task", required=True, choices=sorted(TASKS))
    p.add_argument("--config", default=None, help="a task YAML with the reference's sections")
    p.add_argument("--num-samples", type=int, default=1)
    p.add_argument("--num-videos", type=int, default=1, help="test videos (the reference's test_idx list)")
    p.add_argument("--out", default=None, help="target directory (sample.py's task.target_dir)")
    p.add_argument("--max-batch", type=int, default=None, help="chains per sampler call on a GPU")
    p.add_argument("--seed", type=int, default=37)
    p.add_argument("--annealing-steps", type=int, default=None)
    p.add_argument("--mcmc-steps", type=int, default=None)
    p.add_argument("--diffusion-steps", type=int, default=None)
    p.add_argument("--pixel-space", action="store_true", help="latent_wrapper=False (optimize() on decoded frames)")
    p.add_argument("--noise", default="philox", choices=["philox", "torch"])
    p.add_argument("--prior-solver", default=None, help="override mcmc prior_solver (e.g. gaussian)")
    p.add_argument("--rehearse-on-one-gpu", action="store_true",
                   help="under torchrun on a one-GPU box: all ranks on one card, process group on gloo")
    a = p.parse_args(argv)

    rank, world, device = driver.init_distributed(rehearse_on_one_gpu=a.rehearse_on_one_gpu)
    cfg = load_task(a.task, a.config)
    sc = cfg["sampler"]
    if a.annealing_steps:
        sc["annealing_scheduler_config"]["num_steps"] = a.annealing_steps
    if a.diffusion_steps:
        sc["diffusion_scheduler_config"]["num_steps"] = a.diffusion_steps
    if a.mcmc_steps:
        sc["mcmc_sampler_config"]["num_steps"] = a.mcmc_steps
    if a.prior_solver:
        sc["mcmc_sampler_config"]["prior_solver"] = a.prior_solver
    sc["mcmc_sampler_config"]["noise"] = a.noise
    sc["mcmc_sampler_config"]["nan_guard"] = "chain"         # unrelated chains share a batch here
    op = build_operator(cfg, device, a.config)
    dw = cfg["diffusion_wrapper"]
    model = StandInPrior(dw["video_shape"], dw["latent_shape"]).to(device)
    skw = dict(sc)
    sampler = get_sampler(skw.pop("name"), latent_wrapper=not a.pixel_space, **skw)
    if rank == 0:
        print("task %s on %d rank(s): operator %s, prior = StandInPrior (random weights; the pretrained STeP prior "
              "is a remote download), videos = synthetic, %d sample(s) x %d video(s)"
              % (a.task, world, type(op).__name__, a.num_samples, a.num_videos))
        if getattr(op, "array_file", None):
            print("array file: %s%s" % (op.array_file, " (stand-in: the configured file was not found)"
                                        if op.array_is_standin else ""))
    stats = []
    for idx in range(a.num_videos):
        video = synthetic_video(a.task, idx, dw["video_shape"]).unsqueeze(0).to(device)
        if a.task == "blackhole":
            obs = op.measure(video)                          # restated noise model, seed 24
        else:
            g = torch.Generator(device=device).manual_seed(a.seed + idx)
            clean = op(video)
            obs = clean + op.sigma * torch.randn(clean.shape, device=device, generator=g)
        torch.cuda.synchronize() if device.type == "cuda" else None
        t0 = time.perf_counter()
        samples = driver.sample_posterior(sampler, model, op, obs, a.num_samples, seed=a.seed, device=device,
                                          max_batch=a.max_batch, item_offset=idx * 100003)
        torch.cuda.synchronize() if device.type == "cuda" else None
        dt = time.perf_counter() - t0
        if rank == 0:
            if a.out:
                driver.write_sample_dir(os.path.join(a.out, "%06d" % idx), obs, samples, videos=video)
            # the task's evaluator on every sample (sample.py:71; its wandb summary: max/min/mean/std)
            evaluator = get_evaluator(a.task, op=op)
            main = evaluator.main_eval_fn_name
            with torch.no_grad():
                vals = torch.cat([evaluator(video, obs, samples[i:i + 1].to(device))[main].reshape(-1).float().cpu()
                                  for i in range(samples.shape[0])]).numpy()
            stats.append(dict(video=idx, seconds=dt, samples=a.num_samples,
                              **({"array_file": op.array_file, "array_is_standin": op.array_is_standin}
                                 if getattr(op, "array_file", None) else {}),
                              samples_per_hour=3600.0 * a.num_samples / dt,
                              finite=bool(torch.isfinite(samples).all()),
                              metric=main, metric_max=float(vals.max()), metric_min=float(vals.min()),
                              metric_mean=float(vals.mean()), metric_std=float(vals.std())))
            print(json.dumps(stats[-1]))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return stats


if __name__ == "__main__":
    main()
