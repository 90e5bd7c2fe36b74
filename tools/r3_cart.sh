#!/bin/bash
set -o pipefail
OUT=gpurun_out/${1:-r3cart}; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_shift_uniform.py tests/test_hand_vectors.py -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?; echo "tests exit $rc"; tail -5 $OUT/tests.log | cut -c1-300
[ $rc -ne 0 ] && exit 1
for on in 1; do
  TDR_CART_SKIP=$on timeout -k 10 500 python3 bench.py --config c4 --steps 5 --warmup 2 --no-cpu > $OUT/bench_c4_skip$on.json 2> $OUT/bench_c4_skip$on.err || { tail -5 $OUT/bench_c4_skip$on.err; exit 1; }
  python3 -c "
import json,sys
d=json.load(open('$OUT/bench_c4_skip$on.json'))
print('skip=$on', 'ms/step', d['ms_per_step'], 'score ms', d['roofline']['avg_launch_ms'], 'value', d['value'])"
done
timeout -k 10 300 python3 tools/tune_compact.py c2 "" su > $OUT/tune.txt 2>&1; grep -v "^scene\|amdgpu.ids\|^compact" $OUT/tune.txt
