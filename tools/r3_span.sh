#!/bin/bash
set -o pipefail
OUT=gpurun_out/${1:-r3s}; mkdir -p $OUT; export TMPDIR=/tmp
for k in 16 32 64 128 400; do
  echo "== TDR_SU_SPAN=$k"
  TDR_SU_SPAN=$k timeout -k 10 200 python3 tools/tune_compact.py c2 "${2:-uniform}" su > $OUT/tune_$k.txt 2>&1 || exit 1
  grep -v "^scene\|amdgpu.ids\|^compact" $OUT/tune_$k.txt
done
exit 0
