#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): issue / LDS counters of the MRI step kernel, two passes.
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_sq
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -o run -- python3 $ROOT/bench.py --task mri --batch 64 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/p1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU --output-format csv -d $OUT/p2 -o run -- python3 $ROOT/bench.py --task mri --batch 64 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/p2.log 2>&1
find $OUT -name "*counter_collection.csv"
