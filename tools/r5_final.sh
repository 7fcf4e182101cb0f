#!/bin/bash
# tools/r4_final.sh {a|b|c} — ON THE GPU BOX: the measurement set of round 5's final state -> gpurun_out/r5z_*
ROOT="$PWD"; cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
part="$1"
if [ "$part" = a ]; then
  python3 -m pytest tests -m gpu -q > gpurun_out/r5z_gpu_tests.log 2>&1; tail -2 gpurun_out/r5z_gpu_tests.log
  (time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5z_bench_driver_style.json 2> gpurun_out/r5z_bench_driver_style.err) 2> gpurun_out/r5z_driver_time.txt
  echo "driver-style done"; tail -c 400 gpurun_out/r5z_bench_driver_style.json
  python3 bench.py > gpurun_out/r5z_bench_default.json 2> gpurun_out/r5z_bench_default.err
  echo "default done"
fi
if [ "$part" = b ]; then
  python3 bench.py --quick --steps 16 > /dev/null 2>&1
  bash tools/r5_rocprof_stats.sh > gpurun_out/r5z_rocprof_vs_events.txt 2>&1; cat gpurun_out/r5z_rocprof_vs_events.txt
  python3 bench.py --workload c3d --steps 20 --warmup 5 --no-converge > gpurun_out/r5_c3d_bench.json 2> gpurun_out/r5_c3d_bench.err; echo "c3d line done"
  python3 bench.py --workload terrain --steps 20 --warmup 5 --no-converge > gpurun_out/r5_terrain_bench.json 2> gpurun_out/r5_terrain_bench.err; echo "terrain line done"
fi
if [ "$part" = c ]; then
  for wl in c1 c2 c2p c3 c3d c5 terrain; do
    for st in 20 128; do
      HR_BENCH_TIME_KERNELS=1 python3 bench.py --quick --workload $wl --steps $st $( [ $st = 20 ] && echo "--warmup 5" ) > gpurun_out/wl.json 2>/dev/null && python3 - $wl $st <<'PY'
import json, sys
d = json.load(open("gpurun_out/wl.json")); k = d["extra"]["kernel_ms_rank0"]; n = d["extra"]["kernel_launches_rank0"]
print(f"{sys.argv[1]:8s} {sys.argv[2]:>4s} steps: {d['value']:8.1f} Mrays/s  {d['ms_per_step']:.3f} ms/step  {d['extra']['paths_per_s']/1e6:8.1f} Mpaths/s  rays/path {d['extra']['rays_per_path']:.2f}  trace {k['trace']/max(n['trace'],1):.3f} ms x{n['trace']}  shade {k['shade']/max(n['shade'],1):.3f}")
PY
    done
  done > gpurun_out/r5z_workloads.txt 2>&1
  cat gpurun_out/r5z_workloads.txt
  python3 tools/commit_time.py 2>&1 | tail -1 > gpurun_out/r5z_commit_time.json
  python3 tools/tree_costs.py 2>&1 | grep -v amdgpu > gpurun_out/r5z_tree_costs.txt
  { HR_FUZZ_SEEDS=600 timeout -k 10 500 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -2; echo "HR_FUZZ_SEEDS=600 python -m pytest tests/test_gpu_fuzz.py -m gpu"; timeout -k 10 200 python3 tools/leak_check.py 2>&1 | tail -2; } > gpurun_out/r5z_fuzz_leak.txt 2>&1
  cat gpurun_out/r5z_fuzz_leak.txt
fi
