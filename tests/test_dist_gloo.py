"""CPU, world_size 2 over gloo: the chain-sharded driver (step_amd/driver.py) deals chains to
ranks, runs them batched, and gathers the samples on rank 0 in chain order -- identical to the
single-process result.  (On the GPU node the same code runs one process per GPU over RCCL.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from step_amd import driver
from tests.toy_model import ToyModel


class FakeSampler:
    """Deterministic stand-in for LatentDAPS: output depends only on each chain's start."""

    class _S:
        sigma_max = 100.0

    annealing_scheduler = _S()

    def sample(self, model, z_start, operator, measurement, evaluator=None, **kw):
        return model.decode(z_start / 100.0) + measurement


class NoisySampler(FakeSampler):
    """Draws in-loop noise the way LatentDAPS / MCMCSampler do (chain_randn_like): with the driver's
    per-chain generators a chain's result must not depend on the world size or on max_batch."""

    gens = None

    def set_chains(self, generators=None, seeds=None):
        self.gens = generators

    def sample(self, model, z_start, operator, measurement, evaluator=None, **kw):
        from step_amd.sampler.daps import chain_randn_like
        z = z_start / 100.0
        for _ in range(3):
            z = z + 0.1 * chain_randn_like(z, self.gens)
        return model.decode(z) + measurement


class Model(ToyModel):
    def get_data_shape(self):
        return (3, 1, 8, 8)


def run_single(num_samples, sampler_cls=FakeSampler):
    model = Model((3, 1, 4, 4))
    return driver.sample_posterior(sampler_cls(), model, None, torch.tensor(0.25), num_samples,
                                   seed=37, device=torch.device("cpu"))


def _worker(rank, world, port, num_samples, max_batch, q, sampler_cls=FakeSampler):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, dev = driver.init_distributed("gloo")
    assert (r, w) == (rank, world) and dev.type == "cpu"
    model = Model((3, 1, 4, 4))
    out = driver.sample_posterior(sampler_cls(), model, None, torch.tensor(0.25), num_samples,
                                  seed=37, device=dev, max_batch=max_batch)
    if rank == 0:
        q.put(out)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("num_samples,max_batch,sampler_cls",
                         [(6, None, FakeSampler), (5, 2, FakeSampler), (1, None, FakeSampler),
                          (7, 2, NoisySampler), (4, None, NoisySampler)])
def test_sharded_sampling_equals_single_process(num_samples, max_batch, sampler_cls):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, num_samples, max_batch, q, sampler_cls))
             for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = run_single(num_samples, sampler_cls)
    assert got.shape == (num_samples, 3, 1, 8, 8)
    assert torch.equal(got, want)          # NoisySampler: in-loop noise included, chain by chain


def test_noise_really_enters_and_differs_per_chain():
    out = run_single(3, NoisySampler)
    assert not torch.equal(out, run_single(3, FakeSampler))
    assert not torch.equal(out[0], out[1])


def test_shard_items_partition():
    for n in (0, 1, 7, 64):
        for w in (1, 2, 8):
            parts = [driver.shard_items(n, r, w) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
