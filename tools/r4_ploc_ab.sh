#!/bin/bash
# ON THE GPU BOX: the tree builder A/B (HR_TUNE ploc=0 radix tree only / ploc=1 the cheaper of radix and PLOC (default) / ploc=2 PLOC) per workload
for wl in terrain c3 c2 c3d; do
  for t in "ploc=0" "ploc=1" "ploc=2"; do
    for k in 20 128; do
      HR_TUNE="$t" python bench.py --quick --workload $wl --steps $k --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$wl $t steps $k: %.1f Mrays/s  %.3f ms/step  build %.2f ms  nodes %d' % (d['value'], d['ms_per_step'], d['extra']['bvh_build_ms'], d['config']['bvh_nodes']))"
    done
  done
done
