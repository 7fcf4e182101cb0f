#!/bin/bash
mkdir -p gpurun_out
step() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > gpurun_out/$log 2>&1; local rc=$?; echo "[$log] rc=$rc"; tail -3 gpurun_out/$log | cut -c1-600; if [ $rc -ne 0 ]; then echo "batch ends"; exit 1; fi; }
step 200 r5y_smoke.log python -c "import __graft_entry__ as g; g.smoke()"
step 300 r5y_c5.json python bench.py --quick --parity-seconds 0 --workload c5 --steps 20 --warmup 5
HR_BENCH_FORCE_EXCHANGE=1 step 300 r5y_forced_exchange.json python bench.py --quick --parity-seconds 0 --steps 20 --warmup 5
HR_BENCH_ONE_DEVICE=1 step 400 r5y_rehearsal2.json python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --quick --parity-seconds 0
