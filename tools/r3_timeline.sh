#!/bin/bash
# Runs ON THE GPU BOX: kernel timelines (rocprofv3 --kernel-trace) of the driver-style 20-step run, whole frame and a 1/8 shard.
#   tools/r3_timeline.sh <tag> ["<HR_TUNE>"]   -> gpurun_out/<tag>_tl_{n1,w8}.txt, gpurun_out/<tag>_{n1,w8}.json
tag="$1"; TUNE="$2"
ROOT="$PWD"
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
python3 bench.py --quick --steps 16 > /dev/null 2>&1 || true
for v in n1 w8; do
  extra=""; [ $v = w8 ] && extra="--shard-of 8 --shard-rank 3"
  HR_TUNE="$TUNE" timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "gpurun_out/${tag}_$v" -o t --output-format csv -- python3 bench.py --quick --warmup 5 --no-wakeup --steps 20 $extra > "gpurun_out/${tag}_$v.json" 2> "gpurun_out/${tag}_$v.err" || { echo "trace $v failed"; tail -5 "gpurun_out/${tag}_$v.err"; exit 1; }
  f=$(find "gpurun_out/${tag}_$v" -name '*kernel_trace.csv' | head -1)
  python3 tools/timeline.py "$f" 0 400 > "gpurun_out/${tag}_tl_$v.txt"
  s=$(find "gpurun_out/${tag}_$v" -name '*kernel_stats.csv' | head -1)
  cp "$s" "gpurun_out/${tag}_stats_$v.csv"
  rm -rf "gpurun_out/${tag}_$v"
  HR_TUNE="$TUNE" python3 bench.py --quick --warmup 5 --steps 20 $extra > "gpurun_out/${tag}_plain_$v.json" 2>/dev/null
  python3 - "$v" "gpurun_out/${tag}_plain_$v.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(f"{sys.argv[1]}: {d['value']:.1f} Mrays/s  {d['ms_per_step']:.4f} ms/step  launches {d['extra']['kernel_launches_rank0']}  kernel ms {d['extra']['kernel_ms_rank0']}")
PY
done
