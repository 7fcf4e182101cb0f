#!/bin/bash
# ON THE GPU BOX: threads per workgroup of k_raygen_packets (build_variants/libhrcore_rpN.so, tools/build_variant.sh rpN -DHR_RP_BLOCK=N), batch sizes, k_trace occupancy
run() { # label, env...
  local label="$1"; shift
  for k in 20 128; do
    env "$@" timeout -k 10 120 python bench.py --quick --steps $k --warmup 5 2>/dev/null > gpurun_out/ab_tmp.json || { echo "$label steps $k: FAILED"; return 1; }
    python -c "
import json; d=json.load(open('gpurun_out/ab_tmp.json')); print('$label steps $k: %.1f Mrays/s %.3f ms/step' % (d['value'], d['ms_per_step']))"
  done
}
run "block 256 (default)" X=1 || exit 1
for b in 64 128 512 1024; do run "block $b" HRCORE_LIB=$PWD/build_variants/libhrcore_rp$b.so || exit 1; done
run "batch=32" HR_TUNE=batch=32 || exit 1
run "batch=24" HR_TUNE=batch=24 || exit 1
run "blocks=4" HR_TUNE=blocks=4 || exit 1
run "default again" X=1
