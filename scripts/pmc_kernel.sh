#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): one counter pass over a bench.py configuration, per-kernel
# sums of the largest dispatch.  usage: bash scripts/pmc_kernel.sh <tag> "<counters>" [bench.py args...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; ctr=$2; shift 2
OUT=$ROOT/gpurun_out/pmc_$tag
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d $OUT -o run -- python3 $ROOT/bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-small-batches > $OUT/run.log 2>&1 || exit 1
cd $ROOT
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)[0]
best = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
    v = float(r["Counter_Value"])
    best[k][r["Counter_Name"]] = max(best[k].get(r["Counter_Name"], 0.0), v)
for k, d in best.items():
    if k.startswith("k_"):
        print(k, {n: int(v) for n, v in d.items()})
PY
