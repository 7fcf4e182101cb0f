#!/bin/bash
mkdir -p gpurun_out
run() { local label=$1 k=$2 tune=$3; shift 3; local best=999
  for i in 1 2 3; do
    v=$(HR_TUNE="$tune" timeout -k 10 120 python bench.py --quick --parity-seconds 0 --steps $k --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    best=$(python -c "print(min($best, ${v:-999}))")
  done
  echo "[$label] $k passes, HR_TUNE='$tune' $*: $best ms/step" | tee -a gpurun_out/r5k_tblk.txt
}
S="--shard-of 8 --shard-rank 3"
run shard-memcpy 20 "" $S
run shard-kernel 20 "tblk=1" $S
run n1-memcpy 20 ""
run n1-kernel 20 "tblk=1"
run n1-memcpy 128 ""
run n1-kernel 128 "tblk=1"
run shard-memcpy 128 "" $S
run shard-kernel 128 "tblk=1" $S
ROOT="$PWD"; cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
HR_TUNE="tblk=1" timeout -k 10 200 rocprofv3 --kernel-trace -d "gpurun_out/r5k_w8" -o t --output-format csv -- python3 bench.py --quick --parity-seconds 0 --warmup 5 --steps 20 $S > gpurun_out/r5k_w8.json 2> gpurun_out/r5k_w8.err
f=$(find gpurun_out/r5k_w8 -name '*kernel_trace.csv' | head -1); python3 tools/timeline.py "$f" 0 4000 --all > gpurun_out/r5k_tl_w8.txt; rm -rf gpurun_out/r5k_w8
grep -n "k_raygen_packets" gpurun_out/r5k_tl_w8.txt | tail -2
