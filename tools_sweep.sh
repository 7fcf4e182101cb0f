#!/bin/bash
# usage: tools_sweep.sh "leaf=4,tri=20" "leaf=2,tri=20" ...   (runs bench per HR_TUNE setting, prints one line each)
for t in "$@"; do
  HR_TUNE="$t" python bench.py --cpu-seconds 0 --steps ${STEPS:-16} ${EXTRA} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline'] or {}
print('$t', 'Mrays/s=%.1f'%d['value'], 'ms/step=%.2f'%d['ms_per_step'], 'trace_ms=%.3f'%r.get('avg_launch_ms',0), 'V=%.1f T=%.1f'%(r.get('V',0),r.get('T',0)), 'nodes=',d['config']['bvh_nodes'])"
done
