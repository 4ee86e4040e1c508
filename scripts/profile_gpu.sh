#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel stats + HBM traffic counters (separate passes)
# for both bench tasks (counter passes without the small-batch / sampler-loop extras: one kernel, one size).  Results under gpurun_out/prof/; scripts/pmc_to_json.py folds them into profiles/.
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for task in blackhole mri; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${task}_stats -o run -- python3 $ROOT/bench.py --task $task --steps 100 --warmup 30 --no-cpu-baseline > $OUT/${task}_stats.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${task}_fetch -o run -- python3 $ROOT/bench.py --task $task --steps 3 --warmup 1 --no-cpu-baseline --no-small-batches > $OUT/${task}_fetch.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${task}_write -o run -- python3 $ROOT/bench.py --task $task --steps 3 --warmup 1 --no-cpu-baseline --no-small-batches > $OUT/${task}_write.log 2>&1
done
# matrix-pipe counters of the black-hole DFT kernels (one pass: 6 SQ + 1 GRBM counters)
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_BF16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/blackhole_mfma -o run -- python3 $ROOT/bench.py --task blackhole --steps 3 --warmup 1 --no-cpu-baseline --no-small-batches > $OUT/blackhole_mfma.log 2>&1
find $OUT -name "*.csv" | head -20
