#!/usr/bin/env python3
"""Benchmark of the DAPS measurement-consistency hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--task blackhole|mri] [--batch B]

A "step" is one pass of the hot path over one batch of synthetic chains: for every chain one
Langevin data step = operator loss forward + adjoint (``operator.gradient(x, y, return_loss=True)``,
reference sampler/daps.py:100) at the shapes of configs/task/*.yaml.  The default workload is
BASELINE.json configs[1]: the black-hole operator, 64 x 1 x 256 x 256 video, 8 stations (F = 28,
K_cp = 1344, K_ca = 1280), B = 64 chains per GPU (config 4's num_samples), separable path on the
bf16 matrix cores (exact 3-plane operand split, folded image: step_amd/csrc/bh_dft_sep3.hip).
Inputs are resident in HBM before the timed region.  value = chain-steps / s over all GPUs.

Multi-GPU: chains are independent, so each rank runs its own B chains (weak scaling) with no
data-path collective; the barrier + MAX-over-ranks timing is the only communication.
``python bench.py --gpus N`` (N > 1, no torchrun environment) starts the N ranks itself: the
parent makes no GPU call and runs ``python -m torch.distributed.run --nproc-per-node N bench.py ...``
as a child process, passing its output and exit code on.

Clock ramp: an idle MI355X runs its first ~15 ms of work 20 % below steady state.  Before the W
warm-up steps the bench therefore issues untimed steps until ``--prewarm-ms`` (default 60) of
wall time have passed; the JSON line reports it as ``prewarm_ms``.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MATRIX_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md, HBM3E spec peak


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    # a step is ~0.5 ms: the defaults run long enough to leave the clock ramp of an idle GPU
    # (the first ~15 ms of MFMA work run 20 % slower than steady state) inside the warm-up
    p.add_argument("--steps", type=int, default=400)
    p.add_argument("--warmup", type=int, default=100)
    p.add_argument("--task", default="blackhole", choices=["blackhole", "mri", "mri_radial"])
    p.add_argument("--coils", type=int, default=0,
                   help="--task mri: C > 0 runs the multi-coil operator (no reference counterpart)")
    p.add_argument("--resolution", type=int, default=192,
                   help="--task mri: frame size R x R (192 = the shipped configuration and the tuned kernels; any "
                        "other size runs the any-size kernels, csrc/mri_any.hip)")
    p.add_argument("--batch", type=int, default=None, help="chains per GPU (default: blackhole 64, mri 256)")
    p.add_argument("--path", default="auto", choices=["auto", "sep3", "sep", "dense"])
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-reps", type=int, default=8)
    p.add_argument("--prewarm-ms", type=float, default=60.0,
                   help="untimed steps before the warm-up until this much wall time has passed (clock ramp)")
    p.add_argument("--dry-run", action="store_true",
                   help="rank plumbing only (gloo, no GPU, no kernels): launch, barrier, MAX-reduce, JSON line")
    p.add_argument("--barrier", default="rccl", choices=["gloo", "rccl"],
                   help="what the barriers that bracket the timed region run on (N > 1): the RCCL process group "
                        "(default) or a host-side gloo group.  Measured on one box at world 1 "
                        "(profiles/r03_torchrun_vs_bare.txt): both within 1 %% of the bare run -- each rank's clock "
                        "stops BEFORE the closing barrier either way -- so the default is the one that needs no "
                        "second rendezvous")
    p.add_argument("--rehearse-on-one-gpu", action="store_true",
                   help="N > 1 ranks on ONE card (rank r takes cuda:(r %% visible devices)) with the process group on "
                        "gloo instead of RCCL (which refuses two ranks on a device): the whole torch.distributed.run "
                        "path -- rendezvous, per-rank operators, barriers, MAX, gather, the one JSON line -- except RCCL "
                        "itself; the line says so and is not a scaling measurement")
    p.add_argument("--no-small-batches", action="store_true",
                   help="skip the extra B = 8 / B = 1 and sampler-loop measurements of the default blackhole run")
    return p.parse_args()


def spawn_ranks(n):
    """`python bench.py --gpus N` outside torchrun: start N ranks (one per GPU) as a child
    torch.distributed.run.  This process has made no GPU call and makes none."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


TRAFFIC_SOURCE = [None]


def measured_traffic(kernel, matches):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (newest profiles/r*_traffic.json that has the kernel; FETCH_SIZE/WRITE_SIZE cannot be collected from inside this
    process).  Only returned when the profiled configuration is the one being run; the file it
    came from is reported next to it (`traffic_source`), so that a number measured on an older
    kernel is recognisable as such."""
    if not matches:
        return None
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            with open(path) as fh:
                v = json.load(fh)["kernels"][kernel]["hbm_bytes_per_launch"]
            TRAFFIC_SOURCE[0] = "profiles/" + os.path.basename(path) + " (committed rocprofv3 --pmc passes, not this run)"
            return v
        except (OSError, KeyError, ValueError):
            continue
    return None


def event_ms(pairs):
    return [a.elapsed_time(b) for a, b in pairs]


def build_blackhole(dev, path):
    from step_amd import eht_lite
    from step_amd.forward_operator import BlackHoleImagingTorch
    tab = eht_lite.synthetic_table(8, 64, seed=37, sigma_scale=7e-3)
    op = BlackHoleImagingTorch(imsize=256, w_vis=0, w_amp=0, w_cphase=1, w_logcamp=1, w_flux=0.26,
                               loss_normalize=True, num_frames=64, ref_multiplier=0.001, device=dev,
                               obs_table=tab, dft=path)
    return op, tab


def cpu_baseline_blackhole(tab, reps):
    """The oracle = the reference's formulation (broadcast-multiply DFT evaluated twice per loss,
    generic autograd) on the host cores, B = 1 chain, `reps` gradient calls."""
    from oracle import eht_synth
    from oracle.bh_oracle import BlackholeOracle, closure_tables
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    torch.set_num_threads(threads)
    A = np.stack([eht_synth.ftmatrix_dense(eht_synth.PSIZE, 256, 256, tab["uv"][i]) for i in range(64)])
    A = torch.from_numpy(A).to(torch.complex64).unsqueeze(0)
    cpi, cpc, cai, cac = closure_tables(tab["times"], tab["t1"], tab["t2"], tab["tri"], tab["quad"])
    sigma = torch.from_numpy(tab["sigma"]).float() / 0.001
    orc = BlackholeOracle(A, sigma, torch.tensor(cpi), torch.tensor(cpc), torch.tensor(cai),
                          torch.tensor(cac), 64, 256, 256, w_vis=0, w_amp=0, w_cphase=1,
                          w_logcamp=1, w_flux=0.26, loss_normalize=True)
    g = torch.Generator().manual_seed(37)
    x = torch.rand(1, 64, 1, 256, 256, generator=g) * 2 - 1
    y = orc(torch.rand(1, 64, 1, 256, 256, generator=g) * 2 - 1)
    orc.gradient(x, y, return_loss=True)                       # warm-up
    t0 = time.perf_counter()
    for _ in range(reps):
        orc.gradient(x, y, return_loss=True)
    dt = time.perf_counter() - t0
    out = dict(value=reps / dt, unit="chain-steps/s", cores=threads, kind="port",
               sample="%d x oracle gradient (loss fwd + autograd bwd, reference formulation), "
                      "B=1, 64x1x256x256, F=28" % reps)
    out["sampler"] = cpu_sampler_leg(orc, x, y, reps, "hmc", 0.021)
    return out


def cpu_sampler_leg(orc, x0, y, steps, algo, tau):
    """north_star's 'reference PyTorch-CPU operator AND sampler path': MCMCSampler.sample of the reference
    (oracle/mcmc_oracle.py restates sampler/daps.py:178-208) driving the oracle operator on the same host cores,
    `steps` steps of one call at the shipped algorithm with the 'gaussian' prior solver, pixel-space."""
    from oracle.mcmc_oracle import MCMCOracle
    mc = MCMCOracle(num_steps=steps, lr=1e-9, tau=tau, lr_min_ratio=0.015, prior_solver="gaussian", mc_algo=algo,
                    momentum=0.0)
    xt = x0 + 0.1 * torch.randn(x0.shape, generator=torch.Generator().manual_seed(5))
    t0 = time.perf_counter()
    out = mc.sample(xt, None, x0, orc, y, 0.5, 0.0)
    dt = time.perf_counter() - t0
    return dict(value=x0.shape[0] * steps / dt, unit="chain-steps/s", ms_per_step=1e3 * dt / steps,
                finite=bool(torch.isfinite(out).all()),
                sample="%d steps of the oracle's MCMCSampler.sample (%s, gaussian prior solver, torch noise) on "
                       "the oracle operator, B=%d, pixel-space" % (steps, algo, x0.shape[0]))


def cpu_baseline_mri(reps):
    from oracle.mri_oracle import MRIOracle
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    torch.set_num_threads(threads)
    orc = MRIOracle(6, "same_mask", 192, 12, 24)
    g = torch.Generator().manual_seed(37)
    x = torch.randn(8, 12, 2, 192, 192, generator=g)
    y = orc(torch.randn(1, 12, 2, 192, 192, generator=g))
    orc.gradient(x, y, return_loss=True)
    t0 = time.perf_counter()
    for _ in range(reps):
        orc.gradient(x, y, return_loss=True)
    dt = time.perf_counter() - t0
    out = dict(value=8 * reps / dt, unit="chain-steps/s", cores=threads, kind="port",
               sample="%d x oracle gradient, B=8, 12x2x192x192, acc 6 / 24 ACS" % reps)
    out["sampler"] = cpu_sampler_leg(orc, x, y, reps, "langevin", 0.1)
    return out


class stdout_to_stderr:
    """File descriptor 1 points at stderr inside the block: RCCL prints its version banner on stdout when its
    first communicator comes up, and rank 0's stdout is to carry ONE JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def timed_region(step, steps, sync, barrier, reduce_max, gather):
    """The contract's timed region, shared by the GPU run and --dry-run: barrier + sync, exactly
    `steps` calls of `step`, sync, this rank's clock stops, barrier.  The rank's time is taken BEFORE
    the closing barrier (a collective's own latency is not the workload's); the job's time is the
    MAX over ranks, and every rank's own time is gathered for the JSON line."""
    barrier()
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    sync()
    dt = time.perf_counter() - t0
    barrier()
    return reduce_max(dt), gather(dt)


def dry_run(a, rank, world):
    """The multi-rank plumbing of main() without a GPU (tests/test_bench_spawn.py): rendezvous over
    gloo, the timed region of main() (barriers, per-rank clocks, MAX over ranks, gather), one JSON
    line from rank 0.  A "step" sleeps 1 ms on rank 0, 2 ms on rank 1, ..."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
    comm = Comm(dist if world > 1 else None, None, torch.device("cpu"), world)
    dt, per_rank = timed_region(lambda i: time.sleep(1e-3 * (1 + rank)), a.steps, lambda: None,
                                comm.barrier, comm.reduce_max, comm.gather)
    t = comm.reduce_max(1.0 + rank)
    if rank == 0:
        print(json.dumps(dict(metric="dry run", n_gpus=world, steps=a.steps, warmup=a.warmup,
                              max_over_ranks=t, ms_per_step=1e3 * dt / a.steps,
                              per_rank_ms_per_step=[1e3 * v / a.steps for v in per_rank], dry_run=True)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


class Comm:
    """The only communication of the bench: bracket barriers, MAX and gather of one float per rank.
    `dist` None = single process.  `side` = a host-side (gloo) group for the bracket barriers, or
    None to run them on the default group (RCCL on GPUs)."""

    def __init__(self, dist, side, dev, world):
        self.dist, self.side, self.dev, self.world = dist, side, dev, world

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier(group=self.side) if self.side is not None else self.dist.barrier()

    def reduce_max(self, v):
        if self.dist is None:
            return float(v)
        t = torch.tensor([v], device=self.dev, dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return t.item()

    def gather(self, v):
        if self.dist is None:
            return [float(v)]
        t = torch.tensor([v], device=self.dev, dtype=torch.float64)
        out = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [o.item() for o in out]


def small_batch_line(op, y, dev, B, steps=2000):
    """ms per step and chain-steps/s of the black-hole step at a small batch (wall clock over
    `steps` steps after a warm-up), with the HBM roofline of the step as a whole: the two DFT
    kernels stream 2 x 4*T*H*W bytes per chain and the closure stage rides in them."""
    g = torch.Generator(device=dev).manual_seed(1000 + B)
    x = torch.rand(B, 64, 1, 256, 256, device=dev, generator=g) * 2 - 1
    for _ in range(300):
        op.gradient(x, y, return_loss=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        op.gradient(x, y, return_loss=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    byts = 2 * 4.0 * 64 * 65536 * B
    gbs = byts / (ms * 1e-3) / 1e9
    return dict(ms_per_step=ms, value=B / ms * 1e3, unit="chain-steps/s",
                launches_per_step=2 if op._fused_ok(y, B) else 3,
                roofline=dict(bound="hbm", achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS,
                              algorithmic="whole step: 2 x 4*T*H*W bytes per chain x %d chains / ms_per_step" % B))


def _split_ws_bytes(B, max_rows):
    from step_amd import _lib
    return _lib.lib().mri_loss_grad_split_ws_bytes(B, 12, max_rows, 0)


def small_batch_line_mri(op, y, dev, B, steps=2000):
    """ms per step and chain-steps/s of the MRI step at a small batch, with its HBM roofline on the
    bytes that exist: x read and gradient written per chain, the shared measurement read once."""
    g = torch.Generator(device=dev).manual_seed(1000 + B)
    x = torch.randn(B, 12, 2, 192, 192, device=dev, generator=g)
    for _ in range(300):
        op.gradient(x, y, return_loss=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        op.gradient(x, y, return_loss=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    byts = 16.0 * 12 * 192 * 192 * B + 8.0 * 12 * 192 * 192
    gbs = byts / (ms * 1e-3) / 1e9
    split = bool(y.shape[0] == 1 and op.kernel_variant == 0 and op.split_frames is not False
                 and _split_ws_bytes(B, op._max_rows))
    return dict(ms_per_step=ms, value=B / ms * 1e3, unit="chain-steps/s",
                kernel="mri_loss_grad_split (a frame on 6 or 3 workgroups)" if split else "mri_loss_grad",
                roofline=dict(bound="hbm", achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS,
                              algorithmic="(16*T*R*R x %d chains + 8*T*R*R) bytes / ms_per_step" % B))


def sampler_lines(op, y, dev, B, steps=30, reps=3):
    """ms per step of the loops that call the operator -- MCMCSampler.sample and .optimize
    (sampler/daps.py:178-223) on pixel-space frames, hmc at momentum 0 (the shipped black-hole
    algorithm) with the 'gaussian' prior solver -- so that the driver's run also records what a
    step costs with score, noise and update included.  The step size is one at which the chain
    stays finite on this random start (asserted)."""
    from step_amd.sampler.daps import MCMCSampler
    g = torch.Generator(device=dev).manual_seed(2000 + B)
    x0 = torch.rand(B, 64, 1, 256, 256, device=dev, generator=g) * 2 - 1
    xt = x0 + 0.1 * torch.randn(x0.shape, device=dev, generator=g)
    tau = 0.021
    lr = float(1e-3 * tau ** 2 / op.gradient(x0, y).abs().max())
    out = {}
    for name, noise in (("sample_torch_noise", "torch"), ("sample_philox", "philox"), ("optimize", None)):
        mc = MCMCSampler(num_steps=steps, lr=lr, tau=tau, lr_min_ratio=0.015, prior_solver="gaussian",
                         mc_algo="hmc", momentum=0.0, noise=noise or "torch")
        if noise == "philox":
            mc.set_chains(seeds=torch.arange(37, 37 + B, dtype=torch.int64, device=dev))
        run = (lambda: mc.optimize(x0, op, y, 0.5, 0.0)) if noise is None else \
              (lambda: mc.sample(xt, None, x0, op, y, 0.5, 0.0))
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            res = run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / (reps * steps) * 1e3
        assert torch.isfinite(res).all() and (res != 0).any(), "the chain diverged"
        out[name] = dict(ms_per_step=ms, value=B / ms * 1e3, unit="chain-steps/s")
        del res
    out["config"] = dict(chains=B, steps=steps, algo="hmc, momentum 0", prior_solver="gaussian",
                         note="sample_torch_noise: torch.randn_like per step (the reference's draw), handed to the "
                              "adjoint, which applies the update (langevin_step(eps=...)); sample_philox: update and "
                              "noise inside the adjoint (langevin_step); optimize: update inside the adjoint (descent_step)")
    return out


def sampler_lines_mri(op, y, dev, B, steps=30, reps=3):
    """The MRI counterpart of sampler_lines: MCMCSampler.sample (langevin, the shipped MRI algorithm, torch draws
    handed to the kernel, which applies the update) and .optimize (update inside the kernel) on pixel-space frames
    with the 'gaussian' prior solver."""
    from step_amd.sampler.daps import MCMCSampler
    g = torch.Generator(device=dev).manual_seed(2000 + B)
    x0 = torch.randn(B, 12, 2, 192, 192, device=dev, generator=g)
    xt = x0 + 0.3 * torch.randn(x0.shape, device=dev, generator=g)
    tau = 0.1
    lr = float(1e-3 * tau ** 2 / op.gradient(x0, y).abs().max())
    out = {}
    for name in ("sample_torch_noise", "optimize"):
        mc = MCMCSampler(num_steps=steps, lr=lr, tau=tau, lr_min_ratio=0.011, prior_solver="gaussian",
                         mc_algo="langevin")
        run = (lambda: mc.optimize(x0, op, y, 0.5, 0.0)) if name == "optimize" else \
              (lambda: mc.sample(xt, None, x0, op, y, 0.5, 0.0))
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            res = run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / (reps * steps) * 1e3
        assert torch.isfinite(res).all() and (res != 0).any(), "the chain diverged"
        out[name] = dict(ms_per_step=ms, value=B / ms * 1e3, unit="chain-steps/s")
        del res
    out["config"] = dict(chains=B, steps=steps, algo="langevin", prior_solver="gaussian",
                         note="sample_torch_noise: torch.randn_like per step (the reference's draw), handed to the step "
                              "kernel, which applies the update in its inverse column phase (langevin_step(eps=...)); "
                              "optimize: update inside the kernel (descent_step)")
    return out


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(a.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.dry_run:
        return dry_run(a, rank, world)
    dist = side = None
    gloo_fallback = False
    if "RANK" in os.environ and "MASTER_ADDR" in os.environ:     # launched by torch.distributed.run
        import torch.distributed as dist
        if a.rehearse_on_one_gpu:
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            dist.init_process_group("gloo")
            local = local % max(1, torch.cuda.device_count())
        else:
            # RCCL carries nothing of the workload (chains are independent: no data-path collective), only the
            # bracket barriers and one float per rank.  If it cannot come up on this node (IPC, fabric), say so
            # and run the same bench with those few bytes on gloo instead of failing the whole measurement.
            import datetime
            if local >= torch.cuda.device_count():
                raise SystemExit("bench.py: rank %d wants cuda:%d but %d device(s) are visible (one rank per GPU; "
                                 "--rehearse-on-one-gpu shares a card)" % (rank, local, torch.cuda.device_count()))
            try:
                torch.cuda.set_device(local)
                with stdout_to_stderr():
                    dist.init_process_group("nccl", device_id=torch.device("cuda", local),
                                            timeout=datetime.timedelta(seconds=120))
                    probe = torch.ones(1, device=torch.device("cuda", local))
                    dist.all_reduce(probe)
                    dist.barrier()
                    torch.cuda.synchronize()
                if int(probe.item()) != world:
                    raise RuntimeError("all_reduce of ones gave %r on %d ranks" % (probe.item(), world))
            except Exception as e:
                print("bench.py: RCCL did not come up on rank %d (%s: %s); process group on gloo" %
                      (rank, type(e).__name__, str(e).splitlines()[0][:200] if str(e) else ""), file=sys.stderr)
                try:
                    if dist.is_initialized():
                        dist.destroy_process_group()
                except Exception:
                    pass
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
                dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=120))
                gloo_fallback = True
        if a.barrier == "gloo" and not a.rehearse_on_one_gpu and not gloo_fallback:
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")    # one node: the ranks meet on the loopback
            try:
                side = dist.new_group(backend="gloo")
            except Exception as e:                               # no host-side group: the RCCL barriers do
                print("bench.py: gloo group unavailable (%s); bracket barriers on RCCL" % (e,), file=sys.stderr)
                side = None
        if a.gpus != world and rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; reporting n_gpus=%d" % (a.gpus, world, world),
                  file=sys.stderr)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # blackhole B in {1, 8, 64} (config 4 runs 64 samples), MRI B in {1, 32, 256}
    B = a.batch or (64 if a.task == "blackhole" else 256)
    gen = torch.Generator(device=dev).manual_seed(37 + rank)

    if a.task == "blackhole":
        op, tab = build_blackhole(dev, a.path)
        x = torch.rand(B, 64, 1, 256, 256, device=dev, generator=gen) * 2 - 1
        y = op(torch.rand(1, 64, 1, 256, 256, device=dev, generator=gen) * 2 - 1)
        workload = "blackhole operator loss fwd + adjoint, 64x1x256x256, 8 stations F=28 K_cp=1344 K_ca=1280"
    elif a.task == "mri_radial":
        from step_amd.forward_operator import get_operator
        op = get_operator("mri_radial", spokes=32, resolution=192, frames=12)
        B = a.batch or 64
        x = torch.randn(B, 12, 2, 192, 192, device=dev, generator=gen)
        y = op(torch.randn(1, 12, 2, 192, 192, device=dev, generator=gen))
        workload = ("radial mri operator (gridding NUFFT, no reference counterpart) loss fwd + adjoint, "
                    "12x2x192x192, 32 golden-angle spokes x 192 per frame")
    else:
        from step_amd.forward_operator import get_operator
        R = a.resolution
        cfg = dict(acc_factor=6, mode="same_mask", resolution=R, frames=12, num_acs_lines=R // 8)
        if a.coils > 0:
            op = get_operator("mri_multicoil", coils=a.coils, **cfg)
            B = a.batch or 64
            workload = "multi-coil mri operator (%d coils, no reference counterpart) loss fwd + adjoint, " \
                       "12x2x%dx%d, acc 6 / %d ACS same_mask" % (a.coils, R, R, R // 8)
        else:
            op = get_operator("mri", **cfg)
            if R != 192:
                B = a.batch or 64
            workload = "mri operator loss fwd + adjoint, 12x2x%dx%d, acc 6 / %d ACS same_mask%s" % (
                R, R, R // 8, "" if R == 192 else " (any-size kernels)")
        x = torch.randn(B, 12, 2, R, R, device=dev, generator=gen)
        y = op(torch.randn(1, 12, 2, R, R, device=dev, generator=gen))

    comm = Comm(dist, side, torch.device("cpu") if (a.rehearse_on_one_gpu or gloo_fallback) else dev, world)

    def barrier():
        if dist is not None:
            dist.barrier()                                       # RCCL

    # the first collectives of a process set RCCL up (communicator, kernels: hundreds of ms with the
    # GPU idle): here, ahead of the pre-warm, so that the barrier that opens the timed region takes
    # its 25 us and the clocks stay where the warm-up left them
    for _ in range(3):
        barrier()
        comm.barrier()
    # (the ranks leave that barrier together and the pre-warm below runs by wall time, so they reach
    # the barrier that opens the timed region together too and nobody's GPU idles while waiting)
    # untimed, disclosed pre-warm: leave the clock ramp of an idle GPU before anything is timed
    t_pre = time.perf_counter()
    prewarm_steps = 0
    while (time.perf_counter() - t_pre) * 1e3 < a.prewarm_ms:
        for _ in range(10):
            op.gradient(x, y, return_loss=True)
        torch.cuda.synchronize()
        prewarm_steps += 10
    prewarm_ms = (time.perf_counter() - t_pre) * 1e3
    for _ in range(a.warmup):
        op.gradient(x, y, return_loss=True)
    torch.cuda.synchronize()
    # HIP events (on the launch stream) bracket every kernel of every 4th step of the timed
    # region: each timed event pair costs a few microseconds of queue time, which on all steps
    # would add ~2 % to a 0.45 ms step
    prof = {}
    last = {}

    def step(i):
        op._prof = prof if i % 4 == 0 else None
        last["out"] = op.gradient(x, y, return_loss=True)

    dt, per_rank = timed_region(step, a.steps, torch.cuda.synchronize, comm.barrier, comm.reduce_max, comm.gather)
    op._prof = None
    grad, loss = last["out"]
    assert torch.isfinite(loss).item()

    if rank == 0:
        kern = {k: float(np.mean(event_ms(v))) for k, v in prof.items()}
        dtype = "f32"
        if a.task == "blackhole" and op.path == "sep3":
            dtype = "f32 (3xbf16 exact operand split on the bf16 MFMA, fp32 accumulate)"
        if a.task == "blackhole" and "dft_fwd" not in kern:
            # below 32 chains the step is two launches inside one C call (closure stage in the DFT
            # kernels): no event between them, so the roofline is that of the step as a whole
            ms = kern["fused_step"]
            byts = 2 * 4.0 * 64 * 65536 * B
            gbs = byts / (ms * 1e-3) / 1e9
            roof = dict(bound="hbm", kernel="two-launch step (dft_fwd_sep3 + dft_adj_sep3, closure stage inside)",
                        achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS, traffic=None,
                        algorithmic="whole step: 2 x 4*T*H*W bytes per chain x %d chains" % B, kernels_ms=kern)
        elif a.task == "blackhole":
            T, F, HW = 64, 28, 256 * 256
            flop = 4.0 * T * F * HW * B                       # per direction per launch
            per = {k: dict(ms=kern[k], tflops=flop / (kern[k] * 1e-3) / 1e12) for k in ("dft_fwd", "dft_adj")}
            dom = max(per, key=lambda k: per[k]["ms"])
            if op.path == "sep3":
                # the bf16 matrix work (6 plane products on half the rows) fits under the one pass
                # over the image that each direction must make: the kernel is HBM-bound
                byts = 4.0 * T * HW * B                       # fwd: read x; adj: write grad
                gbs = byts / (per[dom]["ms"] * 1e-3) / 1e9
                roof = dict(bound="hbm", kernel=dom + "_sep3", achieved=gbs, peak=HBM_PEAK_GBS,
                            unit="GB/s", frac=gbs / HBM_PEAK_GBS,
                            traffic=measured_traffic("k_" + dom + "_s3", B == 64),
                            algorithmic="4*T*H*W bytes per chain per direction x %d chains" % B,
                            kernels_ms=kern,
                            # SURVEY 8d prices the DFT in fp32 flops against the fp32 matrix peak:
                            # the same launch in that currency (the kernel does not use that pipe)
                            fp32_mfma_view=dict(achieved=per[dom]["tflops"], peak=FP32_MATRIX_PEAK_TFLOPS,
                                                unit="TFLOP/s", frac=per[dom]["tflops"] / FP32_MATRIX_PEAK_TFLOPS,
                                                algorithmic="4*T*F*H*W flop per chain per direction"),
                            fp32_equivalent_tflops={k: v["tflops"] for k, v in per.items()},
                            other_gbs={k: byts / (v["ms"] * 1e-3) / 1e9 for k, v in per.items() if k != dom})
            else:
                roof = dict(bound="mfma", kernel=dom + "_" + op.path, achieved=per[dom]["tflops"],
                            peak=FP32_MATRIX_PEAK_TFLOPS, unit="TFLOP/s",
                            frac=per[dom]["tflops"] / FP32_MATRIX_PEAK_TFLOPS,
                            traffic=measured_traffic("k_" + dom + "_sep", op.path == "sep" and B == 64),
                            algorithmic="4*T*F*H*W flop per chain per direction x %d chains" % B,
                            kernels_ms=kern,
                            other={k: v["tflops"] for k, v in per.items() if k != dom})
        elif a.task == "mri_radial":
            ms = kern["nufft_loss_grad"]
            byts = (16.0 * 12 * 192 * 192 + 8.0 * 12 * op.J) * B     # read x, write grad, read the samples
            gbs = byts / (ms * 1e-3) / 1e9
            roof = dict(bound="hbm", kernel="nufft step (row FFT, column FFT, interp, spread, column IFFT, row IFFT)",
                        achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS, traffic=None,
                        algorithmic="(16*T*R*R + 8*T*J) bytes per chain x %d chains; the oversampled grids "
                                    "(per frame 8*G*G bytes written and read once each way, plus the half-height "
                                    "intermediate of the pruned transforms) are overhead of gridding" % B,
                        kernels_ms=kern)
        elif a.coils > 0:
            ms = kern["mri_loss_grad_mc"]
            byts = (16.0 + 8.0 * a.coils) * 12 * float(a.resolution) ** 2 * B    # read x, write grad, read y of C coils
            gbs = byts / (ms * 1e-3) / 1e9
            roof = dict(bound="hbm", kernel="mri_loss_grad_mc", achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=gbs / HBM_PEAK_GBS, traffic=None,
                        algorithmic="(16 + 8*C)*T*R*R bytes per chain x %d chains, C = %d" % (B, a.coils),
                        kernels_ms=kern)
        else:
            # One measurement serves every chain (y_batch = 1, the sampler's case), so the bytes that
            # exist are x read + gradient written per chain and y read once: 16*T*R*R*B + 8*T*R*R.
            # That is the primary fraction; SURVEY 8d's 24*T*R*R per chain (a measurement per chain)
            # over-counts y by B - 1 reads and is reported next to it.
            RR = float(a.resolution) ** 2
            exist = 16.0 * 12 * RR * B + 8.0 * 12 * RR
            survey = 24.0 * 12 * RR * B
            ms = kern["mri_loss_grad"]
            gbs = exist / (ms * 1e-3) / 1e9
            roof = dict(bound="hbm", kernel="mri_loss_grad", achieved=gbs, peak=HBM_PEAK_GBS,
                        unit="GB/s", frac=gbs / HBM_PEAK_GBS,
                        traffic=(measured_traffic("k_mri_loss_grad2p", B == 256 and a.resolution == 192)
                                 or measured_traffic("k_mri_loss_grad2", B == 256 and a.resolution == 192)),
                        algorithmic="(16*T*R*R per chain + 8*T*R*R once: one measurement for all chains) x %d chains" % B,
                        survey_8d_view=dict(achieved=survey / (ms * 1e-3) / 1e9, frac=survey / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                            algorithmic="24*T*R*R bytes per chain x %d chains (a measurement per chain)" % B),
                        kernels_ms=kern)
        out = dict(metric="DAPS Langevin steps/sec (fwd+adjoint)", value=world * B * a.steps / dt,
                   unit="chain-steps/s", n_gpus=world, steps=a.steps, warmup=a.warmup,
                   ms_per_step=1e3 * dt / a.steps, higher_is_better=True, scaling="weak",
                   vs_baseline=None, dtype=dtype, data="synthetic",
                   prewarm_ms=prewarm_ms, prewarm_steps=prewarm_steps,
                   per_rank_ms_per_step=[1e3 * v / a.steps for v in per_rank],
                   bracket_barrier=("none (one process)" if dist is None else
                                    "gloo (rehearsal: all ranks on one card, NOT a scaling measurement)"
                                    if a.rehearse_on_one_gpu else
                                    "gloo (RCCL did not come up on this node: see stderr)" if gloo_fallback else
                                    "gloo (host)" if side is not None else "rccl"),
                   config=dict(workload=workload, chains_per_gpu=B, path=getattr(op, "path", "fused"),
                               parallelism="independent chains, %d per GPU" % B),
                   roofline=roof)
        if roof.get("traffic") is not None:
            roof["traffic_source"] = TRAFFIC_SOURCE[0]
        # BASELINE.json's second metric, derived: Langevin steps per posterior sample at the shipped
        # configs; operator work only, the prior network is not in this bench
        per_sample = 1500 if a.task == "blackhole" else 1060
        if a.coils > 0 or a.task == "mri_radial":
            out["config"]["reference_counterpart"] = False
        out["derived"] = dict(samples_per_hour_operator_only=3600.0 * out["value"] / per_sample,
                              langevin_steps_per_sample=per_sample)
        if world == 1 and a.task == "blackhole" and a.batch is None and not a.no_small_batches and op.path == "sep3":
            # the batches a GPU really sees: the reference runs B = 1 (sample.py:64-65) and config 4
            # on 8 GPUs is 64 / 8 = 8 chains per GPU.  Same operator, same step, own roofline.
            out["small_batches"] = {"B%d" % b: small_batch_line(op, y, dev, b) for b in (8, 1)}
            del x
            torch.cuda.empty_cache()
            out["sampler_loop"] = sampler_lines(op, y, dev, B)
        if world == 1 and a.task == "mri" and a.coils == 0 and a.batch is None and not a.no_small_batches \
                and a.resolution == 192:
            # the batches a GPU really sees: the reference runs B = 1 (sample.py:64-65) and config 5
            # on 8 GPUs is 32 / 8 = 4 chains per GPU (frames shared between workgroups there)
            del x
            out["small_batches"] = {"B%d" % b: small_batch_line_mri(op, y, dev, b) for b in (4, 1)}
            torch.cuda.empty_cache()
            out["sampler_loop"] = sampler_lines_mri(op, y, dev, B)
        if world == 1 and not a.no_cpu_baseline:
            if a.task == "blackhole":
                out["cpu_baseline"] = cpu_baseline_blackhole(tab, a.cpu_reps)
            elif a.task == "mri" and a.coils == 0 and a.resolution == 192:
                out["cpu_baseline"] = cpu_baseline_mri(a.cpu_reps * 4)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
