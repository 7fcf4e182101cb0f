#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + separate PMC passes of the default bench.
#   tools/profile.sh <tag> [bench args...]      -> gpurun_out/prof_<tag>/{trace,fetch,write,tcc,sq}/  + bench_<tag>.json
# Afterwards, in the repo:  python tools/summarize_profile.py <tag>   -> profiles/<tag>_*
set -e
tag="$1"; shift
ARGS="--quick --warmup 0 --no-wakeup $*"
ROOT="$PWD"
OUT="$ROOT/gpurun_out/prof_$tag"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
# (a throw-away process first: the first GPU process on a fresh box can run a fifth slower, and the summary compares the HIP-event
# launch averages of the next run with rocprofv3's of the one after)
python3 bench.py --quick --steps 16 > /dev/null 2>&1 || true
python3 bench.py $ARGS > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.err"
echo "[profile] plain bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace --output-format csv -- python3 bench.py $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
echo "[profile] kernel trace done"
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum" "sq:SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  name="${pass%%:*}"; ctrs="${pass#*:}"
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $ctrs -d "$OUT/$name" -o "$name" --output-format csv -- python3 bench.py $ARGS > "$OUT/bench_$name.json" 2> "$OUT/$name.err" || { echo "[profile] pass $name failed"; tail -5 "$OUT/$name.err"; }
  echo "[profile] pmc pass $name done"
done
# keep the merge small: only the csv summaries travel back
find "$OUT" -type f ! -name '*.csv' ! -name '*.json' ! -name '*.err' -delete
find "$OUT" -name '*kernel_trace.csv' -size +20M -delete
du -sh "$OUT"
