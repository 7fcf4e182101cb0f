#!/bin/bash
mkdir -p gpurun_out
step() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > gpurun_out/$log 2>&1; local rc=$?; echo "[$log] rc=$rc"; tail -3 gpurun_out/$log; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT: batch ends"; exit 1; fi; }
step 300 r5h_tail_test.log python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tail_kernel"
step 800 r5h_gpu_suite.log python -m pytest tests -m gpu -x -q
run() { local label=$1 k=$2 tune=$3; shift 3; local best=999
  for i in 1 2 3; do
    v=$(HR_TUNE="$tune" timeout -k 10 120 python bench.py --quick --parity-seconds 0 --steps $k --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    best=$(python -c "print(min($best, ${v:-999}))")
  done
  echo "[$label] $k passes, HR_TUNE='$tune' $*: $best ms/step" | tee -a gpurun_out/r5h_tail_ab.txt
}
S="--shard-of 8 --shard-rank 3"
run shard-notail 20 "tail=0" $S
run shard-tail 20 "" $S
run shard-tail-s2 20 "tailstage=2,tailmax=200000" $S
run shard-tail-s4 20 "tailstage=4" $S
run shard-tail-16k 20 "tailmax=16384" $S
run shard-tail-100k 20 "tailmax=100000" $S
run shard-tail-200k 20 "tailmax=200000" $S
run n1-notail 20 "tail=0"
run n1-tail 20 ""
run n1-tail-200k 20 "tailmax=200000"
run n1-notail 128 "tail=0"
run n1-tail 128 ""
run shard-notail 128 "tail=0" $S
run shard-tail 128 "" $S
HR_TUNE="" timeout -k 10 200 python tools/r4_shard_steps.py > gpurun_out/r5h_shard_steps.txt 2>&1; tail -16 gpurun_out/r5h_shard_steps.txt
