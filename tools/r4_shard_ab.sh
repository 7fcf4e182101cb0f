#!/bin/bash
# tools/r4_shard_ab.sh — runs ON THE GPU BOX: a 1/8 shard (rank 3) of c3 at the driver's 20 passes under scheduling variants
run() { # label, env..., tune
  local label="$1"; shift
  local best=999
  for i in 1 2 3; do
    v=$(env "$@" python bench.py --quick --steps 20 --warmup 5 --shard-of 8 --shard-rank 3 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    best=$(python -c "print(min($best, $v))")
  done
  echo "$label: best of 3 = $best ms/step"
}
run "default" HR_TUNE=""
run "no kernel timing events" HR_TUNE="" HR_BENCH_TIME_KERNELS=0
run "groups=2" HR_TUNE="groups=2"
run "groups=2,blocks=5" HR_TUNE="groups=2,blocks=5"
run "groups=2,batch=10" HR_TUNE="groups=2,batch=10"
run "batch=10" HR_TUNE="batch=10"
run "batch=7" HR_TUNE="batch=7"
run "groups=3" HR_TUNE="groups=3"
run "groups=2 no timing" HR_TUNE="groups=2" HR_BENCH_TIME_KERNELS=0
