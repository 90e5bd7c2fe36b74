#!/bin/bash
# su tests (+ optional timing); stops at the first failing stage so that a faulting kernel is not run twice
set -o pipefail
OUT=gpurun_out/${1:-r3t}; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_shift_uniform.py -m gpu -x -q -k "${2:-shift}" > $OUT/tests.log 2>&1; rc=$?; echo "tests exit $rc"; tail -5 $OUT/tests.log | cut -c1-300
[ $rc -ne 0 ] && exit 1
[ "${3:-tune}" = tune ] && { timeout -k 10 300 python3 tools/tune_compact.py c2 "" su > $OUT/tune.txt 2>&1; grep -v "^scene\|amdgpu.ids\|^compact" $OUT/tune.txt; }
exit 0
