#!/bin/bash
# Runs ON THE GPU BOX: the MRI step at small batches -- wall clock per step (scripts/kbench_mri_small.py)
# and rocprofv3 kernel durations of the same loop.  usage: bash scripts/kprof_mri_small.sh <tag> [B ...]
ROOT=$(pwd); TAG=${1:-run}; shift
OUT=$ROOT/gpurun_out/kprof_mri_${TAG}
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 200 python3 $ROOT/scripts/kbench_mri_small.py "$@" > $OUT/wall.txt 2>&1 || { tail -5 $OUT/wall.txt; exit 1; }
grep "^B=" $OUT/wall.txt
cd /tmp && export TMPDIR=/tmp
for B in "$@"; do
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/B$B -o run -- python3 $ROOT/scripts/kbench_mri_small.py $B > $OUT/log_B$B.txt 2>&1 || exit 1
  f=$(find $OUT/B$B -name "*kernel_stats.csv" | head -1)
  python3 - "$f" $B <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:4]:
    print('B=%s %-50s calls %6s avg %9.1f ns  min %9s' % (sys.argv[2], r['Name'][:50], r['Calls'], float(r['AverageNs']), r['MinNs']))
PY
done
