#!/bin/bash
# usage (ON THE GPU BOX): tools/ab_libs.sh LABEL — the in-tree library (LABEL) against build_variants/libhrcore_prev.so (tools/build_variant.sh prev "" on the
# parent commit) on c3 / c3d / c5 at 20 passes, c3 at 128 and a 1/8 shard, three runs each, then the parity tests: how round 5's kernel experiments were judged
mkdir -p gpurun_out; out=gpurun_out/r5af_$1.txt; rm -f $out
run() { local label=$1 wl=$2 k=$3 lib=$4; shift 4
  for i in 1 2 3; do
    v=$(HR_BENCH_TIME_KERNELS=1 HRCORE_LIB=$lib timeout -k 10 300 python bench.py --quick --parity-seconds 0 --workload $wl --steps $k --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['extra']['kernel_ms_rank0']; print(round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'shade ms', round(k['shade'],3), 'trace ms', round(k['trace'],3))")
    echo "[$label] $wl $k passes $*: $v" >> $out
  done
}
P=$PWD/build_variants/libhrcore_prev.so
for wl in c3 c3d c5; do
  run prev $wl 20 $P
  run $1 $wl 20 ""
done
run prev c3 128 $P
run $1 c3 128 ""
run prev c3 20 $P --shard-of 8 --shard-rank 3
run $1 c3 20 "" --shard-of 8 --shard-rank 3
cat $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r5af_tests.log 2>&1; tail -2 gpurun_out/r5af_tests.log
