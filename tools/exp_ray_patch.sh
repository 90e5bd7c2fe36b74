#!/bin/bash
# the ray-mapped kernel's round-5 changes on / off: bench lines of the polar configurations
#   bash tools/exp_ray_patch.sh "<name>=<v>,..." c2 c5 ...
T=$1; shift
for c in "$@"; do
  echo -n "$T "
  python bench.py --config $c --steps 20 --warmup 5 --no-cpu --tuning $T 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print(d['config']['workload'][:3], 'step %.3f ms' % d['ms_per_step'], 'launch %.3f' % r['avg_launch_ms'], 'init', d['config'].get('init_search_first_step_ms'))"
done
