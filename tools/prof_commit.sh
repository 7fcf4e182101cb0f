#!/bin/bash
# rocprofv3 kernel trace of tools/commit_time.py: which kernels a first commit and a refit spend their time in
ROOT="$PWD"; OUT="$ROOT/gpurun_out/prof_commit"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT" -o commit --output-format csv -- python3 tools/commit_time.py > "$OUT/out.json" 2> "$OUT/err.txt"
python3 - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:16]:
        print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} total {float(r['TotalDurationNs'])/1e6:9.3f} ms avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
find "$OUT" -size +4M -delete
