#!/bin/bash
# Runs ON THE GPU BOX: counter passes over scripts/mri_variant_run.py for kernel variant $1 at batch $2
set -e
ROOT=$(pwd); V=${1:-3}; B=${2:-256}
OUT=$ROOT/gpurun_out/prof_v$V
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() { n=$1; shift; timeout -s KILL 90 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -o run -- python3 $ROOT/scripts/mri_variant_run.py $V $B > $OUT/$n.log 2>&1; }
pass p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pass p2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_IFETCH SQ_INSTS_FLAT
pass p6 SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES
pass p7 SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA
pass p3 FETCH_SIZE
pass p4 WRITE_SIZE
pass p5 TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
python3 - $OUT <<'PY'
import csv, glob, sys
for f in sorted(glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)):
    best = {}
    for r in csv.DictReader(open(f)):
        if 'k_mri_loss_grad' not in r['Kernel_Name']: continue
        best[r['Counter_Name']] = max(best.get(r['Counter_Name'], 0.0), float(r['Counter_Value']))
    for k, v in best.items(): print('%-36s %.4g' % (k, v))
PY
