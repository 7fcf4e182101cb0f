#!/bin/bash
# ON THE GPU BOX: A/B of two builds of the library (HRCORE_LIB), interleaved: tools/r4_lib_ab.sh <variant.so> [workloads...]
V=$1; shift
WLS=${@:-c3}
for wl in $WLS; do
  for rep in 1 2 3; do
    for lib in heatray_amd/csrc/libhrcore.so $V; do
      for k in 20 128; do
        HRCORE_LIB=$PWD/$lib python bench.py --quick --workload $wl --steps $k --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$wl $(basename $lib) steps $k: %.1f Mrays/s  %.3f ms/step  trace avg %.3f ms' % (d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_ms_device_clock') or 0))"
      done
    done
  done
done
