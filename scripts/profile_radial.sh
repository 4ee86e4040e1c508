#!/bin/bash
# Runs ON THE GPU BOX: kernel stats + HBM traffic counters (separate passes) of the radial-MRI step
# (bench.py --task mri_radial, 64 chains).  Results under gpurun_out/prof_radial/.
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_radial
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $ROOT/bench.py --task mri_radial --steps 50 --warmup 10 --no-cpu-baseline > $OUT/stats.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o run -- python3 $ROOT/bench.py --task mri_radial --steps 3 --warmup 1 --no-cpu-baseline > $OUT/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o run -- python3 $ROOT/bench.py --task mri_radial --steps 3 --warmup 1 --no-cpu-baseline > $OUT/write.log 2>&1
ls $OUT/*
