#!/bin/bash
# Runs ON THE GPU BOX: the task runner at the shipped schedules (no shortened steps) and the per-GPU batches of
# BASELINE configs 4 / 5 on 8 GPUs (8 black-hole chains, 4 MRI chains), latent route (stand-in prior through
# LatentWrapper) and pixel-space route; wall time per run.  Output: gpurun_out/run_task_full.txt
set -e
OUT=gpurun_out/run_task_full.txt
: > $OUT
for spec in "blackhole 8" "mri 4"; do
  set -- $spec
  for extra in "" "--pixel-space --prior-solver gaussian"; do
    s=$(date +%s.%N)
    timeout -k 10 500 python -m step_amd.run_task --task $1 --num-samples $2 --max-batch $2 $extra > gpurun_out/rt.log 2>&1
    e=$(date +%s.%N)
    echo "task $1 chains $2 ${extra:-latent route}: $(python3 -c "print('%.1f s' % ($e - $s))") (whole process, incl. imports and table set-up)" >> $OUT
    grep -E "finite|metric|steps|sec|samples" gpurun_out/rt.log | tail -3 >> $OUT
  done
done
cat $OUT
