#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/r5ac.txt
run() { local label=$1 wl=$2 k=$3 lib=$4; shift 4
  for i in 1 2 3; do
    v=$(HR_BENCH_TIME_KERNELS=1 HRCORE_LIB=$lib timeout -k 10 300 python bench.py --quick --parity-seconds 0 --workload $wl --steps $k --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['extra']['kernel_ms_rank0']; print(round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'shade ms', round(k['shade'],3), 'trace ms', round(k['trace'],3))")
    echo "[$label] $wl $k passes $*: $v" >> gpurun_out/r5ac.txt
  done
}
M=$PWD/build_variants/libhrcore_mb3.so
for wl in c3 c3d c5; do for k in 20 128; do
  run 4waves-128vgpr $wl $k ""
  run 3waves-155vgpr $wl $k $M
done; done
run 4waves-128vgpr c3 20 "" --shard-of 8 --shard-rank 3
run 3waves-155vgpr c3 20 $M --shard-of 8 --shard-rank 3
cat gpurun_out/r5ac.txt
