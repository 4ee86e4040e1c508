"""`python bench.py --gpus N` starts its own N ranks (the driver runs exactly that command): the
rank plumbing -- spawn, rendezvous on 127.0.0.1, barrier, MAX over ranks, one JSON line from rank 0
with n_gpus = N -- checked on the CPU with the gloo backend (--dry-run: no GPU, no kernels)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["warmup"] == 1
    assert r["max_over_ranks"] == 2.0            # rank 1's value: the MAX went over both ranks
    # the timed region main() uses: every rank's own clock (stopped before the closing barrier) is
    # in the line, the job's time is their MAX.  A dry-run step sleeps 1 ms on rank 0, 2 ms on rank 1
    per = r["per_rank_ms_per_step"]
    assert len(per) == 2 and r["ms_per_step"] == max(per)
    # lower bounds only: a sleep never returns early, but on a busy host it may return late by any amount
    assert per[0] >= 1.0 and per[1] >= 2.0


def test_gpus_1_stays_in_process():
    r = _run(["--gpus", "1", "--dry-run"])
    assert r["n_gpus"] == 1 and r["max_over_ranks"] == 1.0
    assert len(r["per_rank_ms_per_step"]) == 1 and r["ms_per_step"] == r["per_rank_ms_per_step"][0]


@pytest.mark.gpu
def test_two_ranks_under_torch_distributed_run_on_one_card():
    """The driver's N > 1 launch line on a one-GPU box: `--rehearse-on-one-gpu` puts both ranks on cuda:0 with
    the process group on gloo (RCCL refuses two ranks on a device) -- rendezvous, per-rank operators, bracket
    barriers, MAX over ranks, per-rank times, ONE JSON line from rank 0.  Not a scaling measurement."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "10", "--warmup", "3", "--batch", "8", "--rehearse-on-one-gpu"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 10 and len(d["per_rank_ms_per_step"]) == 2
    assert abs(d["ms_per_step"] - max(d["per_rank_ms_per_step"])) < 1e-9
    assert abs(d["value"] - 2 * 8 * 10 / (d["ms_per_step"] * 10 * 1e-3)) < 1e-3 * d["value"]
    assert "rehearsal" in d["bracket_barrier"] and d["scaling"] == "weak"
