#!/bin/bash
# What bounds score_cart_kernel at config 4?  Separate counter passes over bench.py, summarised by tools/pmc_summary.py.
#   gpurun -- 'bash tools/pmc_cart_bound.sh <tag>'
TAG=${1:-cartbound}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
i=0
for G in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SMEM" \
         "TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" \
         "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); echo "pass $i: $G"
  rocprofv3 --pmc $G -d $OUT/p$i -o pmc --output-format csv -- python3 bench.py --config c4 --steps 2 --warmup 1 --no-cpu > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
done
python3 tools/pmc_summary.py score_cart $(find $OUT -name '*counter_collection.csv' | sort) | tee $OUT/summary.txt
