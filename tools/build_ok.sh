#!/bin/bash
# rebuild libtdr_hip.so; exit non-zero (and say so) when the build failed, so that no GPU run uses a stale library
python3 - "$@" <<'PY'
import sys
import top_down_renderer_amd.build as b
try:
    b.build(force="--force" in sys.argv, extra_flags=[a for a in sys.argv[1:] if a.startswith("-D")])
except Exception as e:
    print("BUILD FAILED:", str(e)[-2000:])
    sys.exit(1)
print("build ok")
PY
