#!/bin/bash
# What bounds a polar scoring kernel of the integer form?  Separate counter passes over tools/time_int_form.py (one
# distribution, one kernel), summarised per kernel by tools/pmc_summary.py.
#   gpurun -- 'bash tools/pmc_ray_bound.sh <tag> <distribution: mix|uniform|gauss5> <ray|su|rayctx> [kernel name filter] [patch=0|1]'
TAG=${1:-raybound}; DIST=${2:-uniform}; AXIS=${3:-ray}; K=${4:-score_polar_ray}; EXTRA=${5:-}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
i=0
for G in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" \
         "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" \
         "TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM" \
         "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  echo "pass $i: $G"; rocprofv3 --pmc $G -d $OUT/p$i -o pmc --output-format csv -- python3 tools/time_int_form.py c2 only=$DIST $AXIS $EXTRA > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
done
python3 tools/pmc_summary.py $K $(find $OUT -name '*counter_collection.csv' | sort) | tee $OUT/summary.txt
