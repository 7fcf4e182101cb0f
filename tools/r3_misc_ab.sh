#!/bin/bash
# GPU box: (1) what the per-kernel HIP-event timing costs the timed region; (2) passes per macro step at 20 and 128 steps
one() { # label, env assignments..., then bench args after --
  label="$1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python3 bench.py --quick "$@" > gpurun_out/misc.json 2> gpurun_out/misc.err || { echo "FAILED $label"; tail -3 gpurun_out/misc.err; return; }
  python3 - "$label" <<'PY'
import json, sys
d = json.load(open("gpurun_out/misc.json")); n = d["extra"]["kernel_launches_rank0"]
print(f"{sys.argv[1]:34s} {d['value']:8.1f} Mrays/s  {d['ms_per_step']:.4f} ms/step  trace launches {n['trace']}")
PY
}
python3 bench.py --quick --steps 16 > /dev/null 2>&1
for rep in 1 2; do
one "20 steps, kernel timing on" HR_TUNE= -- --steps 20 --warmup 5
one "20 steps, kernel timing OFF" HR_BENCH_TIME_KERNELS=0 -- --steps 20 --warmup 5
done
one "128 steps, kernel timing on" HR_TUNE= -- --steps 128
one "128 steps, kernel timing OFF" HR_BENCH_TIME_KERNELS=0 -- --steps 128
for b in 9 10 12 16 20; do
one "20 steps, batch=$b" HR_TUNE=batch=$b -- --steps 20 --warmup 5
done
for b in 9 12 16 24; do
one "128 steps, batch=$b" HR_TUNE=batch=$b -- --steps 128
done
