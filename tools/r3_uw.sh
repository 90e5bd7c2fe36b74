#!/bin/bash
# statistics chains: tests, then timing; stops at the first failing stage
set -o pipefail
OUT=gpurun_out/${1:-r3uw}; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "uw_small or serial_chains or update_weights or prefix or heads_of" > $OUT/tests.log 2>&1; rc=$?; echo "tests exit $rc"; tail -5 $OUT/tests.log | cut -c1-400
[ $rc -ne 0 ] && exit 1
timeout -k 10 120 python3 tools/time_uw_small.py > $OUT/time.txt 2>&1; grep -v amdgpu.ids $OUT/time.txt
if [ "$2" = soak ]; then PYTHONPATH=. timeout -k 10 300 python3 tests/soak_chains.py 400 > $OUT/soak.txt 2>&1; tail -3 $OUT/soak.txt; fi
exit 0
