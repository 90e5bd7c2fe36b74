#!/bin/bash
# same-box A/B of two builds of the library: the previous one (top_down_renderer_amd/libtdr_prev.so) and the current one
set -o pipefail
OUT=gpurun_out/${1:-r3ab}; mkdir -p $OUT; export TMPDIR=/tmp
FILTER=${2:-}
TDR_LIB_PATH=$PWD/top_down_renderer_amd/libtdr_prev.so timeout -k 10 300 python3 tools/tune_compact.py c2 "$FILTER" su > $OUT/tune_prev.txt 2>&1; echo "--- previous build"; grep -v "^scene\|amdgpu.ids\|^compact" $OUT/tune_prev.txt
timeout -k 10 300 python3 tools/tune_compact.py c2 "$FILTER" su > $OUT/tune_cur.txt 2>&1; echo "--- current build"; grep -v "^scene\|amdgpu.ids\|^compact" $OUT/tune_cur.txt
