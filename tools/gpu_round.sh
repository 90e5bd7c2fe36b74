#!/bin/bash
# One GPU-box round: tests, bench (c2), kernel trace and the PMC traffic pass behind bench.py's roofline record.
#   gpurun --timeout 1200 -- 'bash tools/gpu_round.sh <tag> [tests|notests] [configs...]'
set -o pipefail
TAG=${1:-r}; MODE=${2:-tests}; shift 2 || true
CONFIGS=${@:-c2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "$MODE" = tests ]; then
  python -m pytest tests -m gpu -q --durations=10 > $OUT/tests.log 2>&1; echo "tests exit $?" | tee -a $OUT/tests.log
  tail -3 $OUT/tests.log
fi
for C in $CONFIGS; do
  N=$(python3 -c "from top_down_renderer_amd import synth; c=synth.CONFIGS['$C']; print(c.n_particles//8 if c.name in ('c3','c5') else c.n_particles)")
  K=score_polar; [ "$C" = c4 ] && K=score_cart
  STEPS=20; [ "$C" = c4 ] && STEPS=5
  # counter and trace passes: long enough for the span tuner (tdr_su_span_begin: 2 + 5 calls) to settle in the polar configs
  PSTEPS=16; [ "$C" = c4 ] && PSTEPS=3
  # PMC pass first (its own run, counters only), then the record, then the bench proper so that it reports the traffic
  rm -rf $OUT/pmc_$C
  rocprofv3 --pmc TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum -d $OUT/pmc_$C/traffic -o pmc --output-format csv -- python3 bench.py --config $C --steps $PSTEPS --warmup 1 --no-cpu > $OUT/pmc_$C.log 2>&1 || { echo "pmc $C failed"; tail -5 $OUT/pmc_$C.log; }
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $OUT/pmc_$C/issue -o pmc --output-format csv -- python3 bench.py --config $C --steps $PSTEPS --warmup 1 --no-cpu > $OUT/pmc_issue_$C.log 2>&1 || { echo "pmc issue $C failed"; tail -5 $OUT/pmc_issue_$C.log; }
  rocprofv3 --pmc TA_BUSY_avr -d $OUT/pmc_$C/ta -o pmc --output-format csv -- python3 bench.py --config $C --steps $PSTEPS --warmup 1 --no-cpu > $OUT/pmc_ta_$C.log 2>&1 || { echo "pmc ta $C failed"; tail -5 $OUT/pmc_ta_$C.log; }
  rocprofv3 --pmc SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d $OUT/pmc_$C/vmem -o pmc --output-format csv -- python3 bench.py --config $C --steps $PSTEPS --warmup 1 --no-cpu > $OUT/pmc_vmem_$C.log 2>&1 || { echo "pmc vmem $C failed"; tail -5 $OUT/pmc_vmem_$C.log; }
  python3 tools/traffic_from_pmc.py $C $K $N $OUT/pmc_$C $OUT/pmc_${C}_summary.txt || echo "no traffic record for $C"
  cp profiles/score_traffic.json $OUT/score_traffic.json
  CPU=""
  python3 bench.py --config $C --steps $STEPS --warmup 3 $CPU > $OUT/bench_$C.json 2> $OUT/bench_$C.err || { echo "bench $C failed"; tail -5 $OUT/bench_$C.err; }
  cat $OUT/bench_$C.json
  rocprofv3 --kernel-trace --stats -d $OUT/trace_$C -o trace --output-format csv -- python3 bench.py --config $C --steps $PSTEPS --warmup 1 --no-cpu > $OUT/trace_$C.log 2>&1 || echo "trace $C failed"
  F=$(find $OUT/trace_$C -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && cp $F $OUT/kernel_stats_$C.csv && head -8 $F
  rm -rf $OUT/trace_$C/*/*kernel_trace.csv 2>/dev/null
done
