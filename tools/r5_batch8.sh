#!/bin/bash
mkdir -p gpurun_out
step() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > gpurun_out/$log 2>&1; local rc=$?; echo "[$log] rc=$rc"; tail -3 gpurun_out/$log; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT: batch ends"; exit 1; fi; }
export N32=$PWD/build_variants/libhrcore_n32.so
HRCORE_LIB=$N32 step 600 r5l_n32_tests.log python -m pytest tests/test_hit_rule.py tests/test_gpu_parity.py -m gpu -x -q -k "traversal or builders or hostile or deep_tree or hit_rule or phantom or sliver or cornell or soup or terrain or edit or refit or cache or config2 or config3_whole"
step 400 r5l_base_tests.log python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "traversal or cache or edit"
run() { local label=$1 wl=$2 k=$3 lib=$4; local best=0
  for i in 1 2 3; do
    v=$(HRCORE_LIB=$lib timeout -k 10 120 python bench.py --quick --parity-seconds 0 --workload $wl --steps $k --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline'].get('avg_launch_ms_device_clock') or 0)")
    echo "[$label] $wl $k passes: $v" >> gpurun_out/r5l_node32_ab.txt
  done
}
for rep in 1 2; do for wl in c3 c3d terrain c2; do for k in 20 128; do
  run node64 $wl $k ""
  run node32 $wl $k $N32
done; done; done
cat gpurun_out/r5l_node32_ab.txt
