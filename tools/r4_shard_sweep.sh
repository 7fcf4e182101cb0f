#!/bin/bash
# ON THE GPU BOX: knob sweep for a 1/8 shard at 20 passes (best of 3 each)
best() { local b=999; for i in 1 2 3; do v=$(HR_TUNE="$1" python bench.py --quick --steps 20 --warmup 5 --shard-of 8 --shard-rank 3 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"); b=$(python -c "print(min($b, $v))"); done; echo "$1: $b"; }
for t in "" "blocks=4" "blocks=6" "blocks=8" "sdeal=512" "sdeal=1024" "sdeal=128" "refill=8" "refill=24" "fmin=32" "fmax=128" "fprim=128,fgate=4" "heads=4" "heads=6" "sblocks=2" "sblocks=6"; do best "$t"; done
