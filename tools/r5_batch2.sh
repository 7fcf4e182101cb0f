#!/bin/bash
# ON THE GPU BOX (round 5, second batch): the GPU suite with the queue guards, the sorted step log, the memory budget; c3 with the guards; memory under budgets
mkdir -p gpurun_out
step() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > gpurun_out/$log 2>&1; local rc=$?; echo "[$log] rc=$rc"; tail -3 gpurun_out/$log; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT: batch ends"; exit 1; fi; }
step 700 r5c_gpu_suite.log python -m pytest tests -m gpu -x -q
for k in 20 128 20 128; do
  timeout -k 10 120 python bench.py --quick --parity-seconds 0 --workload c3 --steps $k --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('c3 steps $k: %.1f Mrays/s  %.3f ms/step  trace avg %.3f ms' % (d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_ms_device_clock') or 0))" >> gpurun_out/r5c_c3.txt
done
cat gpurun_out/r5c_c3.txt
for b in 0 16 8 4; do timeout -k 10 200 python tools/mem_probe.py c3 512 $b 2>/dev/null >> gpurun_out/r5c_mem.txt; done
cat gpurun_out/r5c_mem.txt
