#!/bin/bash
mkdir -p gpurun_out
step() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > gpurun_out/$log 2>&1; local rc=$?; echo "[$log] rc=$rc"; tail -3 gpurun_out/$log; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT: batch ends"; exit 1; fi; }
step 800 r5q_gpu_suite.log python -m pytest tests -m gpu -x -q
step 400 r5q_soak_vs_oracle.txt python tools/r5_soak_vs_oracle.py c3 640 8
step 300 r5q_soak_digests.txt python tools/r4_soak_digest.py c3 640
step 300 r5q_soak_digests_terrain.txt python tools/r4_soak_digest.py terrain 256
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r5q_bench_driver_style.json 2> gpurun_out/r5q_bench_err.txt; tail -c 600 gpurun_out/r5q_bench_driver_style.json
