#!/bin/bash
# ON THE GPU BOX: camera rays as packets (HR_TUNE packets=1) against the default: per-stage visits, throughput per workload
for t in "packets=0" "packets=1"; do
  echo "== $t: stages of c3"; HR_TUNE="$t" python tools/r3_stage_stats.py c3 2>/dev/null | head -3
done
for wl in ${@:-c3 c3d terrain c2}; do
  for rep in 1 2; do
    for t in "packets=0" "packets=1"; do
      for k in 20 128; do
        HR_TUNE="$t" python bench.py --quick --workload $wl --steps $k --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$wl $t steps $k: %.1f Mrays/s  %.3f ms/step  kernels %s' % (d['value'], d['ms_per_step'], {k: round(v, 2) for k, v in d['extra'].get('kernel_ms_rank0', {}).items()}))"
      done
    done
  done
done
