#!/bin/bash
# ON THE GPU BOX: every dispatch (not only the pipeline's kernels) around the first two macro steps of the timed region, N = 1 and a 1/8 shard
ROOT="$PWD"; cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
python3 bench.py --quick --steps 16 > /dev/null 2>&1 || true
for v in n1 w8; do
  extra=""; [ $v = w8 ] && extra="--shard-of 8 --shard-rank 3"
  timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace -d "gpurun_out/r5j_$v" -o t --output-format csv -- python3 bench.py --quick --parity-seconds 0 --warmup 5 --no-wakeup --steps 20 $extra > "gpurun_out/r5j_$v.json" 2> "gpurun_out/r5j_$v.err" || { echo "trace $v failed"; tail -5 "gpurun_out/r5j_$v.err"; }
  f=$(find "gpurun_out/r5j_$v" -name '*kernel_trace.csv' | head -1)
  python3 tools/timeline.py "$f" 0 4000 --all > "gpurun_out/r5j_tl_$v.txt"
  m=$(find "gpurun_out/r5j_$v" -name '*memory_copy_trace.csv' | head -1)
  [ -n "$m" ] && cp "$m" "gpurun_out/r5j_memcpy_$v.csv"
  rm -rf "gpurun_out/r5j_$v"
done
