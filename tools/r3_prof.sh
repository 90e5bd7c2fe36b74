#!/bin/bash
# kernel-level time of one scoring call (tools/tune_compact.py, one distribution, one kernel)
set -o pipefail
TAG=${1:-r3prof}; DIST=${2:-100% Gaussian 5}; AXIS=${3:-su-only}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python3 tools/tune_compact.py c2 "$DIST" $AXIS > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; }
F=$(find $OUT/trace -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && cp $F $OUT/kernel_stats.csv && head -14 $F | cut -c1-170
rm -rf $OUT/trace
