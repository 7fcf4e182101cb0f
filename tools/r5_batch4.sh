#!/bin/bash
# ON THE GPU BOX: how a 1/8 shard's 20 passes (one partly filled batch) are best dealt out over injecting steps, with the packet kernel beside k_trace
mkdir -p gpurun_out
run() { # <label> <passes> <HR_TUNE> <bench args...>
  local label=$1 k=$2 tune=$3; shift 3
  local best=999
  for i in 1 2 3; do
    v=$(HR_TUNE="$tune" timeout -k 10 120 python bench.py --quick --parity-seconds 0 --steps $k --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    best=$(python -c "print(min($best, ${v:-999}))")
  done
  echo "[$label] $k passes, HR_TUNE='$tune' $*: $best ms/step" | tee -a gpurun_out/r5f_burst_sweep.txt
}
S="--shard-of 8 --shard-rank 3"
run shard-default 20 "" $S
run shard-bsplit16 20 "bsplit=16" $S
run shard-bsplit12 20 "bsplit=12" $S
run shard-bsplit10 20 "bsplit=10" $S
run shard-bsplit8 20 "bsplit=8" $S
run shard-bsplit18 20 "bsplit=18" $S
run shard-batch16 20 "batch=16" $S
run shard-default 128 "" $S
run shard-batch16 128 "batch=16" $S
run n1-default 20 ""
run n1-batch8 20 "batch=8"
run n1-batch10 20 "batch=11"
run n1-default 128 ""
S4="--shard-of 4 --shard-rank 1"
run shard4-default 20 "" $S4
run shard4-bsplit10 20 "bsplit=10" $S4
run shard4-bsplit12 20 "bsplit=12" $S4
S2="--shard-of 2 --shard-rank 1"
run shard2-default 20 "" $S2
run shard2-bsplit10 20 "bsplit=10" $S2
