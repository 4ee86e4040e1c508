#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel stats of an end-to-end task run (stand-in prior).
# usage: bash scripts/prof_run_task.sh <tag> [run_task args...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
OUT=$ROOT/gpurun_out/prof_rt_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 -m step_amd.run_task "$@" > $OUT/run.log 2>&1 || exit 1
cd $ROOT
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print("%-90s %7s %9.1f us  %5.1f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print("GPU busy total %.2f s" % (tot / 1e9))
PY
tail -1 $OUT/run.log
