#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of the black-hole step at batch $1 (FUSED=0/1 in env)
ROOT=$(pwd); B=${1:-64}; TAG=${2:-run}
OUT=$ROOT/gpurun_out/kprof_${TAG}_B$B
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 $ROOT/scripts/kbench_small.py $B > $OUT/log.txt 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:6]:
    print('%-60s calls %6s avg %9.1f ns  min %9s' % (r['Name'][:60], r['Calls'], float(r['AverageNs']), r['MinNs']))
PY
