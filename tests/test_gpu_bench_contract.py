"""`python bench.py` on the GPU prints ONE JSON line with the contract's keys (metric, value, unit,
n_gpus, steps, warmup, ms_per_step, higher_is_better, scaling, vs_baseline, dtype, data, config with
a `workload`) plus `roofline` -- for every task the bench knows.  Short runs: the numbers are not
checked, their presence, types and mutual consistency are."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("args,chains", [(["--batch", "4"], 4),
                                         (["--batch", "33"], 33),
                                         (["--path", "sep", "--batch", "2"], 2),
                                         (["--path", "dense", "--batch", "2"], 2),
                                         (["--task", "mri", "--coils", "1", "--batch", "2"], 2),
                                         (["--task", "mri", "--batch", "8"], 8),
                                         (["--task", "mri", "--coils", "3", "--batch", "2"], 2),
                                         (["--task", "mri_radial", "--batch", "2"], 2)])
def test_bench_line(args, chains, dev):
    r = run(args + ["--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["steps"] == 3 and r["warmup"] == 1 and r["higher_is_better"] is True
    assert r["scaling"] == "weak" and r["vs_baseline"] is None and r["data"] == "synthetic"
    assert "workload" in r["config"] and "model" not in r["config"]
    assert r["value"] == pytest.approx(chains / (r["ms_per_step"] * 1e-3), rel=1e-6)      # chain-steps/s, whole job
    roof = r["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s")
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], rel=1e-6) and 0 < roof["frac"] < 1
