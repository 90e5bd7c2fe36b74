#!/bin/bash
# one short line per config: ms per step, scoring launch, its two kernels, first update with the init search
#   bash tools/bench_line.sh c2 c5 c3 ...
for c in "$@"; do
  S=20; [ "$c" = c4 ] && S=5
  python bench.py --config $c --steps $S --warmup 5 --no-cpu 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; s = r['shares']
print(d['config']['workload'][:3], 'step %.3f ms' % d['ms_per_step'], 'launch %.3f' % r['avg_launch_ms'],
      ('dense %.3f scattered %.3f (%d)' % (s['dense_ms'], s['scattered_ms'], s['scattered_particles'])) if s else '',
      'init %s' % d['config'].get('init_search_first_step_ms'))"
done
