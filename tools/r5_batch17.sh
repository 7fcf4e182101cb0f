#!/bin/bash
mkdir -p gpurun_out
step() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > gpurun_out/$log 2>&1; local rc=$?; echo "[$log] rc=$rc"; tail -5 gpurun_out/$log; if [ $rc -ne 0 ]; then echo "batch ends"; exit 1; fi; }
step 300 r5x_tables_tests.log python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tables or blue_noise or qmc or radial or pentagon or table_kinds"
step 300 r5x_tables_time.log python tools/tables_time.py
step 800 r5x_gpu_suite.log python -m pytest tests -m gpu -x -q
