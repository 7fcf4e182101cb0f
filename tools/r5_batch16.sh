#!/bin/bash
mkdir -p gpurun_out
step() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > gpurun_out/$log 2>&1; local rc=$?; echo "[$log] rc=$rc"; tail -3 gpurun_out/$log; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT: batch ends"; exit 1; fi; }
step 800 r5w_gpu_suite.log python -m pytest tests -m gpu -x -q
run() { local label=$1 wl=$2 k=$3 lib=$4; shift 4
  for i in 1 2 3; do
    v=$(HR_BENCH_TIME_KERNELS=1 HRCORE_LIB=$lib timeout -k 10 300 python bench.py --quick --parity-seconds 0 --workload $wl --steps $k --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['extra']['kernel_ms_rank0']; print(round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'raygen ms', round(k['raygen'],3))")
    echo "[$label] $wl $k passes $*: $v" >> gpurun_out/r5w_pk32.txt
  done
}
P=$PWD/build_variants/libhrcore_prev.so
for wl in c3 terrain c2 c5 c3d; do for k in 20 128; do
  run pk-node4 $wl $k $P
  run pk-node32 $wl $k ""
done; done
run pk-node4 c3 20 $P --shard-of 8 --shard-rank 3
run pk-node32 c3 20 "" --shard-of 8 --shard-rank 3
cat gpurun_out/r5w_pk32.txt
