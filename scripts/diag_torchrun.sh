#!/bin/bash
# bare `bench.py --gpus 1` against `torchrun --nproc-per-node 1 bench.py` on ONE box, alternating,
# with the bracket barriers on the host (gloo) and on RCCL, and round 2's bench.py for reference.
# usage (GPU box): bash scripts/diag_torchrun.sh > gpurun_out/diag_torchrun.log
set -e
cd "$(dirname "$0")/.."
A="--steps 20 --warmup 5 --no-small-batches --no-cpu-baseline"
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517"
pick() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['roofline'].get('kernels_ms', {})
        print('$1', 'ms_per_step %.4f' % d['ms_per_step'], 'value %.0f' % d['value'], {a: round(b, 4) for a, b in k.items()}, d.get('bracket_barrier'))
"; }
for rep in 1 2 3; do
  timeout -k 10 120 python bench.py --gpus 1 $A 2>/dev/null | pick bare
  timeout -k 10 120 $TR bench.py --gpus 1 $A --barrier gloo 2>/dev/null | pick torchrun_gloo
  timeout -k 10 120 $TR bench.py --gpus 1 $A --barrier rccl 2>/dev/null | pick torchrun_rccl
  timeout -k 10 120 $TR scripts/probes/bench_r02.py --gpus 1 $A 2>/dev/null | pick torchrun_r02
done
