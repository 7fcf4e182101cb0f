#!/bin/bash
# GPU box: what the per-kernel timing events cost (HR_BENCH_TIME_KERNELS=0 switches them off; the bench line then has no kernel times)
for a in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --shard-of 8" "--steps 128 --warmup 12"; do
  echo "== $a"
  for rep in 1 2; do
    for tk in 1 0; do
      HR_BENCH_TIME_KERNELS=$tk python3 bench.py --quick $a > gpurun_out/tc.json 2>/dev/null || { echo failed; exit 1; }
      python3 - $tk <<'PY'
import json, sys
d = json.load(open("gpurun_out/tc.json"))
print(f"time_kernels={sys.argv[1]}  {d['value']:8.1f} Mrays/s  {d['ms_per_step']:.4f} ms/step")
PY
    done
  done
done
