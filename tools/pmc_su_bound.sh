#!/bin/bash
# What bounds the shift-uniform scoring kernel for a given particle distribution?  Separate counter passes over
# tools/tune_compact.py (one distribution, one kernel), summarised per kernel by tools/pmc_summary.py.
#   gpurun -- 'bash tools/pmc_su_bound.sh <tag> "<distribution name filter>" [su-only|lane-only]'
TAG=${1:-subound}; DIST=${2:-bench mix}; AXIS=${3:-su-only}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
i=0
for G in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
         "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" \
         "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" \
         "SQ_INSTS_BRANCH SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_IFETCH" \
         "TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" \
         "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" \
         "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_ICACHE_MISSES"; do
  i=$((i+1))
  echo "pass $i: $G"; rocprofv3 --pmc $G -d $OUT/p$i -o pmc --output-format csv -- python3 tools/tune_compact.py c2 "$DIST" $AXIS > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
done
python3 tools/pmc_summary.py score_polar $(find $OUT -name '*counter_collection.csv' | sort) | tee $OUT/summary.txt
