#!/bin/bash
# What bounds the 40-rotation search at config 5?  Separate counter passes over bench.py (one timed step is enough: the
# search runs in the first update), summarised per kernel by tools/pmc_summary.py.
#   gpurun -- 'bash tools/pmc_init_bound.sh <tag> [kernel name filter]'
TAG=${1:-initbound}; K=${2:-score_init}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
i=0
for G in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" \
         "TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" \
         "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); echo "pass $i: $G"
  rocprofv3 --pmc $G -d $OUT/p$i -o pmc --output-format csv -- python3 bench.py --config c5 --steps 1 --warmup 0 --no-cpu > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
done
python3 tools/pmc_summary.py $K $(find $OUT -name '*counter_collection.csv' | sort) | tee $OUT/summary.txt
