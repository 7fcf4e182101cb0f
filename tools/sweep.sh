#!/bin/bash
# usage: tools/sweep.sh "leaf=4,tri=20" "leaf=2,tri=20" ...   (runs bench per HR_TUNE setting, prints one line each)
for t in "$@"; do
  HR_TUNE="$t" python bench.py --cpu-seconds 0 --no-pmc --no-converge --steps ${STEPS:-16} ${EXTRA} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline'] or {}
g=d['extra'].get('gpu_traversal_counters') or {}
print('$t', 'Mrays/s=%.1f'%d['value'], 'ms/step=%.2f'%d['ms_per_step'], 'kernel_ms', {k: round(v,2) for k,v in d['extra']['kernel_ms_rank0'].items()}, 'V4=%.1f T=%.1f'%(g.get('node4_visits_per_closest_ray',0),g.get('tri_tests_per_closest_ray',0)), 'nodes=',d['config']['bvh_nodes'])"
done
