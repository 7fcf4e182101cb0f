#!/bin/bash
mkdir -p gpurun_out
run() { local label=$1 wl=$2 k=$3 lib=$4
  for i in 1 2 3; do
    v=$(HR_BENCH_TIME_KERNELS=1 HRCORE_LIB=$lib timeout -k 10 300 python bench.py --quick --parity-seconds 0 --workload $wl --steps $k --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['extra']['kernel_ms_rank0']; print(round(d['value'],1), 'shade ms', round(k['shade'],3))")
    echo "[$label] $wl $k passes: $v" >> gpurun_out/r5s_shade_occupancy.txt
  done
}
for wl in c3 c3d; do for k in 20 128; do
  run minblocks4 $wl $k ""
  run minblocks5 $wl $k $PWD/build_variants/libhrcore_hit5.so
  run minblocks6 $wl $k $PWD/build_variants/libhrcore_hit6.so
done; done
cat gpurun_out/r5s_shade_occupancy.txt
