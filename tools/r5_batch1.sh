#!/bin/bash
# ON THE GPU BOX (round 5, first batch): the hit rule's tests, the whole GPU suite, the A/B of the rule's three forms in k_trace, the long
# GPU-vs-oracle render.  A step that hits its time limit ends the batch (nothing else is started on a GPU that may be wedged).
mkdir -p gpurun_out
step() { # step <seconds> <log> <cmd...>
  local lim=$1 log=$2; shift 2
  timeout -k 10 $lim "$@" > gpurun_out/$log 2>&1; local rc=$?
  echo "[$log] rc=$rc"; tail -2 gpurun_out/$log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT: batch ends"; exit 1; fi
}
step 300 r5b_hit_rule.log python -m pytest tests/test_hit_rule.py -m gpu -x -q
step 600 r5b_gpu_suite.log python -m pytest tests -m gpu -x -q
for rep in 1 2 3; do
  for lib in heatray_amd/csrc/libhrcore.so build_variants/libhrcore_hb0.so build_variants/libhrcore_hb2.so; do
    for k in 20 128; do
      HRCORE_LIB=$PWD/$lib timeout -k 10 120 python bench.py --quick --parity-seconds 0 --workload c3 --steps $k --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('c3 $(basename $lib) steps $k: %.1f Mrays/s  %.3f ms/step  trace avg %.3f ms' % (d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_ms_device_clock') or 0))" >> gpurun_out/r5b_hitbox_ab.txt
    done
  done
done
cat gpurun_out/r5b_hitbox_ab.txt
step 400 r5b_soak_after.txt python tools/r5_soak_vs_oracle.py c3 640 8
step 300 r5b_soak_digests.txt python tools/r4_soak_digest.py c3 640
