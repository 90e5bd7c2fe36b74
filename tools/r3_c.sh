#!/bin/bash
set -o pipefail
OUT=gpurun_out/${1:-r3c}; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_shift_uniform.py tests/test_gpu_parity.py -m gpu -x -q -k "shift_uniform or compact" > $OUT/tests_su.log 2>&1; echo "tests exit $?" | tee -a $OUT/tests_su.log
tail -5 $OUT/tests_su.log
timeout -k 10 300 python3 tools/tune_compact.py c2 "" su > $OUT/tune_su.txt 2>&1; cat $OUT/tune_su.txt
