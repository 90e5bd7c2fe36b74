#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3b; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 300 python3 tools/tune_compact.py c2 "" su > $OUT/tune_su.txt 2>&1; cat $OUT/tune_su.txt
timeout -k 10 600 bash tools/pmc_su_bound.sh r3b_pmc_su "bench mix" su-only > $OUT/pmc_su.log 2>&1; tail -60 $OUT/pmc_su.log
