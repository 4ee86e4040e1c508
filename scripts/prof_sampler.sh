#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel stats of the pixel-space sampler loop.
# usage: bash scripts/prof_sampler.sh <tag> [bench_sampler.py args...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
OUT=$ROOT/gpurun_out/prof_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 $ROOT/scripts/bench_sampler.py "$@" > $OUT/run.log 2>&1 || exit 1
cd $ROOT
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print("%-100s %6s %10.1f us" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
tail -1 $OUT/run.log
