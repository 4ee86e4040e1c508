#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel stats of one bench.py configuration.
# usage: bash scripts/prof_bench.sh <tag> [bench.py args...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
OUT=$ROOT/gpurun_out/prof_b_$tag
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 $ROOT/bench.py "$@" --no-cpu-baseline > $OUT/run.log 2>&1 || exit 1
cd $ROOT
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:12]:
    print("%-80s %6s %9.1f us %5.1f %%" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
tail -c 600 $OUT/run.log
