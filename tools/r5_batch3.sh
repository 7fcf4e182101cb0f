#!/bin/bash
# ON THE GPU BOX (round 5, third batch): counter list; GPU suite; memory under budgets; a first look at the 1/8 shard's batch size with packets beside k_trace
mkdir -p gpurun_out
step() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > gpurun_out/$log 2>&1; local rc=$?; echo "[$log] rc=$rc"; tail -3 gpurun_out/$log; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT: batch ends"; exit 1; fi; }
(cd /tmp && TMPDIR=/tmp timeout -k 10 120 rocprofv3 -L > $OLDPWD/gpurun_out/r5e_counters.txt 2>&1); grep -c "" gpurun_out/r5e_counters.txt
step 700 r5e_gpu_suite.log python -m pytest tests -m gpu -x -q
for b in 0 16 8; do timeout -k 10 200 python tools/mem_probe.py c3 512 $b 2>/dev/null >> gpurun_out/r5e_mem.txt; done
cat gpurun_out/r5e_mem.txt
shard() { # <label> <HR_TUNE>
  local best=999
  for i in 1 2 3; do
    v=$(HR_TUNE="$2" timeout -k 10 120 python bench.py --quick --parity-seconds 0 --steps 20 --warmup 5 --shard-of 8 --shard-rank 3 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    best=$(python -c "print(min($best, ${v:-999}))")
  done
  echo "W=8 rank 3, 20 passes, [$1] HR_TUNE='$2': $best ms/step" | tee -a gpurun_out/r5e_shard_sweep.txt
}
shard default ""
shard batch16 "batch=16"
shard batch10 "batch=10"
shard batch8 "batch=8"
shard batch5 "batch=5"
shard batch4 "batch=4"
shard batch8-corun2 "batch=8,corun=2"
shard batch8-blocks "batch=8,cblocks=4"
shard nopackets "packets=0"
