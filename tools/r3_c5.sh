#!/bin/bash
set -o pipefail
OUT=gpurun_out/${1:-r3c5}; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_shift_uniform.py -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?; echo "tests exit $rc"; tail -3 $OUT/tests.log | cut -c1-300
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python3 tools/tune_compact.py c2 "" su > $OUT/tune.txt 2>&1; grep -v "^scene\|amdgpu.ids\|^compact" $OUT/tune.txt
timeout -k 10 500 python3 bench.py --config c5 --steps 8 --warmup 2 --no-cpu > $OUT/bench_c5.json 2> $OUT/bench_c5.err || { tail -5 $OUT/bench_c5.err; exit 1; }
python3 -c "
import json
d=json.load(open('$OUT/bench_c5.json'))
print('c5 ms/step', d['ms_per_step'], 'score ms', d['roofline']['avg_launch_ms'], 'value', d['value'], 'init ms', d['config'].get('init_search_first_step_ms'))"
