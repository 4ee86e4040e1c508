#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel stats + HBM traffic counters (separate passes)
# for both bench tasks.  Results under gpurun_out/prof/; scripts/pmc_to_json.py folds them into profiles/.
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for task in blackhole mri; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${task}_stats -o run -- python3 $ROOT/bench.py --task $task --steps 100 --warmup 30 --no-cpu-baseline > $OUT/${task}_stats.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${task}_fetch -o run -- python3 $ROOT/bench.py --task $task --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${task}_fetch.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${task}_write -o run -- python3 $ROOT/bench.py --task $task --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${task}_write.log 2>&1
done
find $OUT -name "*.csv" | head -20
