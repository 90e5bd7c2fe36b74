#!/bin/bash
# occupancy sensitivity of the shift-uniform kernel: dynamic LDS padding limits workgroups (= waves per SIMD) per CU
for pad in 0 8192 13000 21000 34000; do
  echo "pad $pad"
  python bench.py --config c2 --steps 20 --warmup 5 --no-cpu --tuning su_lds_pad=$pad 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; s = r['shares']
print('step %.3f ms' % d['ms_per_step'], 'launch %.3f' % r['avg_launch_ms'], 'dense %.3f scattered %.3f' % (s['dense_ms'], s['scattered_ms']))"
done
