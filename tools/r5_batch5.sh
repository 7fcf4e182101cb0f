#!/bin/bash
mkdir -p gpurun_out
run() { local label=$1 k=$2 tune=$3; shift 3; local best=999
  for i in 1 2 3; do
    v=$(HR_TUNE="$tune" timeout -k 10 120 python bench.py --quick --parity-seconds 0 --steps $k --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    best=$(python -c "print(min($best, ${v:-999}))")
  done
  echo "[$label] $k passes, HR_TUNE='$tune' $*: $best ms/step" | tee -a gpurun_out/r5g_pmin.txt
}
S="--shard-of 8 --shard-rank 3"
run shard-default 20 "" $S
run shard-bsplit16 20 "bsplit=16" $S
run shard-pmin16 20 "pmin=16" $S
run shard-bsplit16-pmin16 20 "bsplit=16,pmin=16" $S
run shard-bsplit16-pmin8 20 "bsplit=16,pmin=8" $S
run n1-default 20 ""
run n1-pmin16 20 "pmin=16"
run n1-pmin8 20 "pmin=8"
for t in "" "bsplit=16"; do echo "== HR_TUNE=$t" >> gpurun_out/r5g_shard_steps.txt; HR_TUNE="$t" timeout -k 10 200 python tools/r4_shard_steps.py >> gpurun_out/r5g_shard_steps.txt 2>&1; done
cat gpurun_out/r5g_shard_steps.txt | tail -40
for wl in c2 c3 c3d terrain c5; do for m in 1 2; do
  echo "== $wl sprobe=$m" >> gpurun_out/r5g_shadow_probe.txt
  HR_TUNE="sprobe=$m" timeout -k 10 200 python bench.py --quick --parity-seconds 0 --workload $wl --steps 20 --warmup 0 2>&1 | grep "shadow probe" | tail -1 >> gpurun_out/r5g_shadow_probe.txt
done; done
cat gpurun_out/r5g_shadow_probe.txt
