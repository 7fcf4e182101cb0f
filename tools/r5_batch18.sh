#!/bin/bash
mkdir -p gpurun_out
HR_DEBUG_PIPE=1 timeout -k 10 300 python bench.py --quick --parity-seconds 0 --workload c3 --steps 20 --warmup 5 --shard-of 8 --shard-rank 3 > gpurun_out/r5y_grow_w8.json 2> gpurun_out/r5y_grow_w8.err
grep -n "grow\|growths" gpurun_out/r5y_grow_w8.err | tail -30
HR_DEBUG_PIPE=1 timeout -k 10 300 python bench.py --quick --parity-seconds 0 --workload c3 --steps 20 --warmup 5 > gpurun_out/r5y_grow_n1.json 2> gpurun_out/r5y_grow_n1.err
grep -n "grow\|growths" gpurun_out/r5y_grow_n1.err | tail -30
