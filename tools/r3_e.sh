#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
TDR_SU_PROTO=1 timeout -k 10 600 bash tools/pmc_su_bound.sh r3e_pmc_proto "100% Gaussian 5" su-only > gpurun_out/r3e.log 2>&1; tail -45 gpurun_out/r3e.log
