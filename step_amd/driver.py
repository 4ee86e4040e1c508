"""Chain-sharded posterior sampling driver: the multi-GPU counterpart of the reference's
sequential loop (reference sample.py:56-76: per test video, ``num_samples`` chains one after
another at batch size 1 on one GPU).

Posterior chains are independent, so the path shards with no data-path collective:
work items (video index, sample index) are dealt round-robin to ranks, each rank batches its
chains of one video into ONE ``sampler.sample`` call (the operator kernels and LatentDAPS are
batch-agnostic; the measurement broadcasts from batch 1), and the only communication is a
gather of the finished samples to rank 0 (RCCL over xGMI when the backend is nccl; gloo on
CPU in the tests).  Per-chain seeds are ``seed + global chain id``: the reference draws all
chains from one global stream (sample.py:28), which no sharded run can reproduce, so
chain-level parity with the reference is statistical while operator-level parity is exact.
"""
import os

import torch
import torch.distributed as dist

__all__ = ["init_distributed", "shard_items", "chain_generator", "gather_to_rank0",
           "sample_posterior", "write_sample_dir"]


def init_distributed(backend=None, rehearse_on_one_gpu=False):
    """One process per GPU (torchrun env).  Returns (rank, world, device).
    ``rehearse_on_one_gpu``: every rank takes cuda:(LOCAL_RANK % visible devices) and the process group runs on
    gloo (RCCL refuses two ranks on a device) -- the sharded path on a one-GPU box, RCCL itself excepted."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    cuda = torch.cuda.is_available()
    if rehearse_on_one_gpu:
        backend = "gloo"
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        if cuda:
            local = local % max(1, torch.cuda.device_count())
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(backend or ("nccl" if cuda else "gloo"))
    if cuda:
        torch.cuda.set_device(local)
        return rank, world, torch.device("cuda", local)
    return rank, world, torch.device("cpu")


def shard_items(n_items, rank, world):
    """Item ids of this rank: r, r + P, r + 2P, ...  (SURVEY section 8e)."""
    return list(range(rank, n_items, world))


def chain_generator(seed, chain_id, device):
    g = torch.Generator(device=device)
    g.manual_seed(int(seed) + int(chain_id))
    return g


def gather_to_rank0(local, ids, n_items, group=None):
    """local: [n_local, ...] tensor of this rank's results for item ids `ids`.
    Returns on rank 0 a tensor [n_items, ...] in item order, None elsewhere.  Ranks may hold
    different counts (n_items need not divide the world size): every rank pads to the maximum."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        out = torch.empty((n_items,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        out[torch.as_tensor(ids, dtype=torch.long, device=local.device)] = local
        return out
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n_max = (n_items + world - 1) // world
    # a gloo group gathers host tensors only: device results cross it through the host (rehearsal, CPU tests)
    via_host = local.is_cuda and dist.get_backend(group) == "gloo"
    wire = torch.device("cpu") if via_host else local.device
    pad = torch.zeros((n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=wire)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, bufs, dst=0, group=group)
    if rank != 0:
        return None
    out = torch.empty((n_items,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        rid = shard_items(n_items, r, world)
        out[torch.as_tensor(rid, dtype=torch.long, device=local.device)] = bufs[r][: len(rid)].to(local.device)
    return out


def sample_posterior(sampler, model, operator, measurement, num_samples, seed=37, device=None,
                     max_batch=None, evaluator=None, gt=None, group=None, item_offset=0):
    """Draw `num_samples` posterior samples for ONE measurement, chains sharded over ranks and
    batched per rank.  Returns [num_samples, *video_shape] on rank 0 (None on other ranks).

    sampler: LatentDAPS-like (get_start / sample); model: prior with get_latent_shape().
    Every random draw of chain c -- start noise, HMC velocity, Langevin noise, re-noising -- comes
    from that chain's own generator / Philox seed (``seed + item_offset + c``) when the sampler has
    ``set_chains`` (LatentDAPS does): a chain's result is then independent of the world size, of
    `max_batch` and of which chains share its batch.  `item_offset` separates measurements
    (test videos) that reuse one base seed."""
    rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    device = device or next(model.parameters()).device
    ids = shard_items(num_samples, rank, world)
    outs = []
    step = max_batch or max(len(ids), 1)
    for i in range(0, len(ids), step):
        chunk = ids[i: i + step]
        gens = [chain_generator(seed, item_offset + c, device) for c in chunk]
        z = torch.stack([torch.randn(*model.get_latent_shape(), device=device, generator=g) for g in gens])
        if hasattr(sampler, "set_chains"):
            seeds = torch.tensor([int(seed) + item_offset + c for c in chunk], dtype=torch.int64, device=device)
            sampler.set_chains(generators=gens, seeds=seeds)
        else:
            # a sampler without per-chain randomness: its in-loop noise comes from the default
            # generator, seeded per chunk (reproducible for a fixed sharding only)
            (torch.cuda.manual_seed if device.type == "cuda" else torch.manual_seed)(
                int(seed) + 100003 * (item_offset + chunk[0] + 1))
        z_start = z * sampler.annealing_scheduler.sigma_max
        kw = dict(gt=gt) if gt is not None else {}
        # detached: the final decode of a prior with trainable weights would otherwise hand on its autograd graph
        outs.append(sampler.sample(model, z_start, operator, measurement, evaluator=evaluator, **kw).detach())
        if hasattr(sampler, "set_chains"):
            sampler.set_chains(None, None)
    if outs:
        local = torch.cat(outs, dim=0)
    else:                      # more ranks than chains: this rank contributes an empty shard
        local = torch.empty((0,) + tuple(model.get_data_shape()), device=device)
    return gather_to_rank0(local, ids, num_samples, group)


def write_sample_dir(idx_dir, observations, samples, videos=None):
    """On-disk layout of the reference's loop (sample.py:59-67) for one test video: ``observations.pth``
    (the measurement container, under the reference's pickle identity when it is a black-hole
    observation), ``videos.pth`` (ground truth, optional) and one ``sample_{i}.pth`` per posterior
    sample, each ``[1, *video_shape]`` on the CPU as the reference writes them.  Rank 0 only."""
    from pathlib import Path

    from .forward_operator.blackhole import BlackHoleObservation, save_observation
    idx_dir = Path(idx_dir)
    idx_dir.mkdir(parents=True, exist_ok=True)
    if isinstance(observations, BlackHoleObservation):
        save_observation(observations, idx_dir / "observations.pth")
    else:
        torch.save(observations.to("cpu"), idx_dir / "observations.pth")
    if videos is not None:
        torch.save(videos.to("cpu"), idx_dir / "videos.pth")
    for i in range(samples.shape[0]):
        torch.save(samples[i:i + 1].detach().to("cpu").clone(), idx_dir / ("sample_%d.pth" % i))
