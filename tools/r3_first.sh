#!/bin/bash
# round 3, first GPU call: the shift-uniform kernel's parity tests, an A/B bench at config 2, the instruction-cost table
set -o pipefail
OUT=gpurun_out/r3a; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_shift_uniform.py -m gpu -x -q --durations=5 > $OUT/tests_su.log 2>&1; echo "tests exit $?" | tee -a $OUT/tests_su.log
tail -15 $OUT/tests_su.log
TDR_SHIFT_UNIFORM=0 timeout -k 10 200 python3 bench.py --config c2 --steps 20 --warmup 3 --no-cpu > $OUT/bench_c2_su0.json 2> $OUT/bench_c2_su0.err; cat $OUT/bench_c2_su0.json | cut -c1-400
timeout -k 10 200 python3 bench.py --config c2 --steps 20 --warmup 3 --no-cpu > $OUT/bench_c2_su1.json 2> $OUT/bench_c2_su1.err; cat $OUT/bench_c2_su1.json | cut -c1-400
timeout -k 10 120 tools/_bin/valu_cost 8 > $OUT/valu_cost_8.txt 2>&1; cat $OUT/valu_cost_8.txt
timeout -k 10 120 tools/_bin/valu_cost 2 > $OUT/valu_cost_2.txt 2>&1
