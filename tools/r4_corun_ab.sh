#!/bin/bash
# ON THE GPU BOX: the packet kernel of a step beside k_trace on a second stream (HR_TUNE corun=1); k_trace's workgroups per CU in steps with
# such a kernel beside it (cblocks) and in the others (blocks)
for t in ${@:-"corun=0" "corun=0,blocks=4" "corun=1,blocks=5,cblocks=3" "corun=1,blocks=4,cblocks=3" "corun=1,blocks=4,cblocks=2" "corun=1,blocks=5,cblocks=2" "corun=1,blocks=4,cblocks=4"}; do
  for k in 20 128; do
    HR_TUNE="$t" timeout -k 10 120 python bench.py --quick --workload ${WL:-c3} --steps $k --warmup 5 2>/dev/null > gpurun_out/corun_tmp.json || { echo "$t steps $k: FAILED"; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/corun_tmp.json')); print('${WL:-c3} $t steps $k: %.1f Mrays/s %.3f ms/step' % (d['value'], d['ms_per_step']))"
  done
done
