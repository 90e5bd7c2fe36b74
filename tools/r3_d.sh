#!/bin/bash
set -o pipefail
OUT=gpurun_out/${1:-r3d}; mkdir -p $OUT; export TMPDIR=/tmp
TDR_SU_PROTO=1 timeout -k 10 300 python3 tools/tune_compact.py c2 "" su-only > $OUT/tune_proto.txt 2>&1; cat $OUT/tune_proto.txt
