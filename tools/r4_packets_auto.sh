#!/bin/bash
# ON THE GPU BOX: the packet selector per workload: the union factor its probe measures, its decision, and throughput against packets=0 / 1
for wl in ${@:-c1 c2 c3 c3d terrain c5}; do
  echo "== $wl: $(HR_DEBUG_PIPE=1 python bench.py --quick --workload $wl --steps 20 --warmup 5 2>&1 >/dev/null | grep 'packet probe' | sort | uniq -c | sort -rn | head -3 | tr '\n' ';')"
  for t in "packets=0" "packets=1" "packets=2"; do
    for k in 20 128; do
      HR_TUNE="$t" python bench.py --quick --workload $wl --steps $k --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$wl $t steps $k: %.1f Mrays/s  %.3f ms/step' % (d['value'], d['ms_per_step']))"
    done
  done
done
