#!/bin/bash
# TIMING EXPERIMENT: how fast would score_polar_su_kernel be if a non-empty bin's class-plane cell came out of
# LDS instead of a 64-lane global gather?  Builds a variant library whose generated loop reads a (meaningless) LDS halfword at
# the place of the gather — wrong weights, right instruction mix — and times config 2 / 3 / 5 with both libraries.
#   here:      bash tools/exp_lds_gather.sh build      (writes top_down_renderer_amd/libtdr_hip_exp.so, restores the header)
#   GPU box:   bash tools/exp_lds_gather.sh run
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  python3 tools/gen_su_asm.py --experiment-lds-gather
  python3 -c "
from top_down_renderer_amd import build as b
b.OUT = b.OUT.replace('libtdr_hip.so', 'libtdr_hip_exp.so'); b.OBJ_DIR += '_exp'
print(b.build(force=True))"
  python3 tools/gen_su_asm.py
  git diff --stat -- top_down_renderer_amd/csrc/tdr_score_su_asm.h
else
  for LIB in "" "$PWD/top_down_renderer_amd/libtdr_hip_exp.so"; do
    echo "== ${LIB:-the product library}"
    TDR_LIB_PATH=$LIB bash tools/bench_line.sh c2 c3 c5
  done
fi
