#!/bin/bash
# tools/r4_shard_ratio.sh [steps] — ON THE GPU BOX: N = 1 and every rank's shard of an 8-way split of c3, back to back on one device (best of 3 each)
K=${1:-20}
best() { # args of bench.py
  local b=999
  for i in 1 2 3; do
    v=$(python bench.py --quick --parity-seconds 0 --steps $K --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    b=$(python -c "print(min($b, $v))")
  done
  echo $b
}
n1=$(best)
echo "N=1: $n1 ms/step"
worst=0
for r in 0 1 2 3 4 5 6 7; do
  v=$(best --shard-of 8 --shard-rank $r)
  echo "W=8 rank $r: $v ms/step"
  worst=$(python -c "print(max($worst, $v))")
done
python -c "print(f'W=8 max over ranks {$worst:.4f} ms/step; N=1 {$n1:.4f}; ratio {$n1/$worst:.3f} x at $K passes')"
