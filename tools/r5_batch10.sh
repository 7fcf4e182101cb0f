#!/bin/bash
mkdir -p gpurun_out
step() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > gpurun_out/$log 2>&1; local rc=$?; echo "[$log] rc=$rc"; tail -3 gpurun_out/$log; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT: batch ends"; exit 1; fi; }
export N64=$PWD/build_variants/libhrcore_n64.so
step 600 r5n_tests.log python -m pytest tests/test_hit_rule.py tests/test_gpu_parity.py -m gpu -x -q -k "traversal or builders or hostile or deep_tree or hit_rule or phantom or sliver or cornell or soup or terrain or edit or refit or cache or config2 or config3_whole or large_scene"
run() { local label=$1 wl=$2 k=$3 lib=$4
  for i in 1 2 3; do
    v=$(HRCORE_LIB=$lib timeout -k 10 300 python bench.py --quick --parity-seconds 0 --workload $wl --steps $k --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")
    echo "[$label] $wl $k passes: $v" >> gpurun_out/r5n_node32v2_workloads.txt
  done
}
for wl in terrain c3 c3d c2; do for k in 20 128; do
  run node64 $wl $k $N64
  run node32v2 $wl $k ""
done; done
cat gpurun_out/r5n_node32v2_workloads.txt
for lib in "$N64" ""; do
  HRCORE_LIB=$lib timeout -k 10 300 python bench.py --workload terrain --steps 20 --warmup 5 --no-pmc --no-converge --cpu-seconds 0 --parity-seconds 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('terrain counters [${lib:-node32v2}]:', d['extra']['gpu_traversal_counters'])"
done
