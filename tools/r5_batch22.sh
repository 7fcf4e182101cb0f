#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/r5ae.txt
run() { local label=$1 wl=$2 k=$3 lib=$4; shift 4
  for i in 1 2 3; do
    v=$(HR_BENCH_TIME_KERNELS=1 HRCORE_LIB=$lib timeout -k 10 300 python bench.py --quick --parity-seconds 0 --workload $wl --steps $k --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['extra']['kernel_ms_rank0']; print(round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'shade ms', round(k['shade'],3), 'trace ms', round(k['trace'],3))")
    echo "[$label] $wl $k passes $*: $v" >> gpurun_out/r5ae.txt
  done
}
B=$PWD/build_variants
for wl in c3 c5; do for k in 20; do
  run inline $wl $k ""
  for n in tex texenv texsamp all; do run $n $wl $k $B/libhrcore_$n.so; done
done; done
cat gpurun_out/r5ae.txt
