#!/bin/bash
# Runs ON THE GPU BOX: one rocprofv3 --pmc pass per argument (a quoted counter list), default bench with 16 steps;
# prints the per-launch mean of every counter for k_trace and k_shade.    tools/pmc.sh "A B" "C D E" ...
ROOT="$PWD"; OUT="$ROOT/gpurun_out/pmc"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
i=0
for ctrs in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs -d "$OUT/p$i" -o p --output-format csv -- python3 bench.py --quick --warmup 0 --no-wakeup --steps ${STEPS:-16} ${EXTRA} > "$OUT/p$i.json" 2> "$OUT/p$i.err" || { echo "pass $i ($ctrs) failed"; tail -3 "$OUT/p$i.err"; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
acc = {}
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hr::", "").split("<")[0]
        if k not in ("k_trace", "k_shade"):
            continue
        acc.setdefault(k, {}).setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
        acc[k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for k in sorted(acc):
    for c in sorted(acc[k]):
        v = list(acc[k][c].values())
        print(f"{k:8s} {c:40s} mean/launch {sum(v)/len(v):18.1f}   max {max(v):18.1f}   n={len(v)}")
PY
find "$OUT" -type f -size +2M -delete
