#!/bin/bash
# GPU box: the measurement set of round 3's final state -> gpurun_out/r3z_*
set -x
ROOT="$PWD"; cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
tools/profile.sh r3z --steps 20 > gpurun_out/r3z_profile.log 2>&1
tools/profile.sh r3z128 --steps 128 > gpurun_out/r3z128_profile.log 2>&1
(time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3z_bench_driver_style.json 2> gpurun_out/r3z_bench_driver_style.err) 2> gpurun_out/r3z_driver_time.txt
echo "driver-style done"
for wl in c1 c2 c2p c5 terrain; do
  for st in 20 128; do
    python3 bench.py --quick --workload $wl --steps $st $( [ $st = 20 ] && echo "--warmup 5" ) > gpurun_out/wl.json 2>/dev/null && python3 - $wl $st <<'PY'
import json, sys
d = json.load(open("gpurun_out/wl.json")); k = d["extra"]["kernel_ms_rank0"]; n = d["extra"]["kernel_launches_rank0"]
print(f"{sys.argv[1]:8s} {sys.argv[2]:>4s} steps: {d['value']:8.1f} Mrays/s  {d['ms_per_step']:.3f} ms/step  rays/path {d['extra']['rays_per_path']:.2f}  trace {k['trace']/max(n['trace'],1):.3f} ms x{n['trace']}  shade {k['shade']/max(n['shade'],1):.3f}")
PY
  done
done > gpurun_out/r3z_workloads.txt 2>&1
echo "workloads done"
for w in 1 2 4 8; do echo "== W=$w steps=20"; tools/shards.sh $w 20 "" | tail -1; done > gpurun_out/r3z_shards.txt 2>&1
for w in 1 2 4 8; do echo "== W=$w steps=128"; tools/shards.sh $w 128 "" | tail -1; done >> gpurun_out/r3z_shards.txt 2>&1
echo "shards done"
python3 tools/commit_time.py > gpurun_out/r3z_commit_time.txt 2>&1
tail -5 gpurun_out/r3z_commit_time.txt
# fuzz campaign, leak check, throughput of the three estimators
{
  echo "== round 3 final build: fuzz campaign, leak check, estimator throughput (c3, 64 steps)"
  HR_FUZZ_SEEDS=600 timeout -k 10 400 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -2
  echo "HR_FUZZ_SEEDS=600 python -m pytest tests/test_gpu_fuzz.py -m gpu"
  timeout -k 10 200 python3 tools/leak_check.py 2>&1 | tail -2
  for est in reference env_mis all_lights; do
    python3 bench.py --quick --steps 64 --warmup 12 --estimator $est > gpurun_out/est.json 2>/dev/null && python3 - $est <<'PY'
import json, sys
d = json.load(open("gpurun_out/est.json"))
print(f"{sys.argv[1]} {d['value']:.1f} Mrays/s {d['ms_per_step']:.3f} ms/pass {d['extra']['rays_per_path']:.2f} rays/path")
PY
  done
} > gpurun_out/r3z_fuzz_leak_estimators.txt 2>&1
echo "fuzz/leak/estimators done"
