#!/bin/bash
# tools/pmc_variant.sh <tag> <lib or ""> "<counters>" <bench args...> : one rocprofv3 --pmc pass, per-kernel sums
tag="$1"; lib="$2"; ctrs="$3"; shift 3
ROOT="$PWD"; OUT="$ROOT/gpurun_out/pmcv_$tag"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
HRCORE_LIB="$lib" timeout -k 10 240 rocprofv3 --kernel-trace --pmc $ctrs -d "$OUT" -o p --output-format csv -- python3 bench.py --quick --warmup 0 --no-wakeup "$@" > "$OUT/bench.json" 2> "$OUT/err.txt" || { echo "pass failed"; tail -3 "$OUT/err.txt"; }
python3 - "$OUT" "$tag" <<'PY'
import csv, glob, sys, json
acc = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hr::", "").split("<")[0]
        acc.setdefault(k, {}).setdefault(r["Counter_Name"], 0.0)
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
json.dump(acc, open(sys.argv[1] + "/sums.json", "w"), indent=1)
for k in ("k_shade", "k_trace"):
    if k in acc:
        print(sys.argv[2], k, {c: f"{v:.4g}" for c, v in acc[k].items()})
PY
find "$OUT" -name "*.csv" -size +2M -delete
