#!/bin/bash
# tools/shards.sh W STEPS "<HR_TUNE>"  — renders every rank's shard of a W-way split, one after the other, on one GPU
# (emulation: no collective); prints ms/step per rank and the max, which bounds the N = W job.
W="$1"; STEPS="$2"; TUNE="$3"
for r in $(seq 0 $((W-1))); do
  HR_TUNE="$TUNE" python bench.py --quick --parity-seconds 0 --warmup 5 --steps $STEPS --shard-of $W --shard-rank $r > gpurun_out/sh.json 2> gpurun_out/sh.err || { echo "rank $r failed"; tail -3 gpurun_out/sh.err; exit 1; }
  python - $r <<'PY'
import json, sys
d = json.load(open("gpurun_out/sh.json"))
print(f"rank {sys.argv[1]}: {d['ms_per_step']:.4f} ms/step  rays {d['extra']['rays']:.0f}")
PY
done | tee gpurun_out/sh.txt
python - "$TUNE" <<'PY'
import re, sys
ms = [float(re.search(r": ([0-9.]+) ms", l).group(1)) for l in open("gpurun_out/sh.txt")]
rays = [float(re.search(r"rays ([0-9.]+)", l).group(1)) for l in open("gpurun_out/sh.txt")]
print(f"[{sys.argv[1]}] max {max(ms):.4f} ms/step, mean {sum(ms)/len(ms):.4f}, total rays {sum(rays):.0f}")
PY
