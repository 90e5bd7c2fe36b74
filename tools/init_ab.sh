#!/bin/bash
# A/B of the 40-rotation search at config 5 (250 000 particles without a heading): on-the-fly f16 split vs pre-split half
# records, record loads in flight, XCD order.   gpurun -- 'bash tools/init_ab.sh <tag> [notests]'
set -o pipefail
TAG=${1:-init}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
if [ "$2" != notests ]; then
python -m pytest tests/test_gpu_fuzz.py -q -k "init_search" > $OUT/tests.log 2>&1; echo "tests exit $?" | tee -a $OUT/tests.log; tail -3 $OUT/tests.log
python -m pytest tests/test_gpu_parity.py -q -k "c5 or init_search" >> $OUT/tests.log 2>&1; echo "tests exit $?" | tee -a $OUT/tests.log; tail -3 $OUT/tests.log
fi
# (the library reads no environment variables: bench.py --tuning name=value,... calls tdr_config_tuning)
for V in "init_ahead=1" "init_ahead=2" "init_ahead=3"; do
  echo "== $V"
  python3 bench.py --tuning "$V" --config c5 --steps 5 --warmup 1 --no-cpu 2> "$OUT/bench_$V.err" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('cold', d['config'].get('init_search_first_step_cold_ms'), 'warm', d['config'].get('init_search_first_step_ms'), 'steady', d['ms_per_step'])" || tail -5 "$OUT/bench_$V.err"
done
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python3 bench.py --config c5 --steps 3 --warmup 1 --no-cpu > $OUT/trace.log 2>&1 || echo "trace failed"
F=$(find $OUT/trace -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && cp $F $OUT/kernel_stats_c5.csv && head -6 $F | cut -c1-150
rm -rf $OUT/trace/*/*kernel_trace.csv 2>/dev/null
