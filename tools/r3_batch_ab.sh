#!/bin/bash
# GPU box: passes per macro step (HR_TUNE batch=N), driver-style 20 steps and 128 steps, two rounds
one() { label="$1"; tune="$2"; shift 2
  HR_TUNE="$tune" python3 bench.py --quick "$@" > gpurun_out/misc.json 2> gpurun_out/misc.err || { echo "FAILED $label"; tail -3 gpurun_out/misc.err; return; }
  python3 - "$label" <<'PY'
import json, sys
d = json.load(open("gpurun_out/misc.json")); n = d["extra"]["kernel_launches_rank0"]
print(f"{sys.argv[1]:34s} {d['value']:8.1f} Mrays/s  {d['ms_per_step']:.4f} ms/step  trace launches {n['trace']}")
PY
}
python3 bench.py --quick --steps 16 > /dev/null 2>&1
for rep in 1 2; do
for b in 9 11 12 13 14; do one "20 steps, batch=$b" "batch=$b" --steps 20 --warmup 5; done
done
for rep in 1 2; do
for b in 9 12 14; do one "128 steps, batch=$b" "batch=$b" --steps 128; done
done
for b in 9 12; do one "32 steps, batch=$b" "batch=$b" --steps 32 --warmup 2; one "40 steps, batch=$b" "batch=$b" --steps 40 --warmup 2; one "c5 32 steps, batch=$b (x4 px)" "batch=$b" --steps 32 --workload c5; done
