#!/bin/bash
# ON THE GPU BOX: a 1/8 shard at 20 passes: (A) round 3's timing (HIP events around every kernel) vs (B) k_trace's own clock, no events; and N = 1 both ways
one() { env "$@" python bench.py --quick --steps 20 --warmup 5 $ARGS 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for ARGS in "--shard-of 8 --shard-rank 3" ""; do
  for i in 1 2 3; do
    a=$(one HRCORE_LIB=$PWD/build_variants/libhrcore_r4base.so HRCORE_ALLOW_OLD_ABI=1 HR_BENCH_TIME_KERNELS=1)
    b=$(one HR_BENCH_TIME_KERNELS=0)
    c=$(one HR_BENCH_TIME_KERNELS=1)
    echo "[$ARGS] base+events $a   clock,no-events $b   clock+events $c"
  done
done
