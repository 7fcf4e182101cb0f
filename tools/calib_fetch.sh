#!/bin/bash
# Runs on the GPU box: builds tools/calib_fetch.hip and reads FETCH_SIZE for its three kernels.
set -e
ROOT="$PWD"; OUT="$ROOT/gpurun_out/calib_fetch"; rm -rf "$OUT"; mkdir -p "$OUT"
hipcc --offload-arch=gfx950 -O3 -o "$OUT/calib_fetch" tools/calib_fetch.hip 2>/dev/null
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
"$OUT/calib_fetch" > "$OUT/plain.txt"
cat "$OUT/plain.txt"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o calib --output-format csv -- "$OUT/calib_fetch" > /dev/null 2> "$OUT/fetch.err"
python3 - "$OUT" <<'PY'
import csv, glob, sys, json
out = sys.argv[1]
rows = {}
for f in glob.glob(out + "/fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            rows.setdefault(r["Kernel_Name"].split("(")[0], {}).setdefault(r["Dispatch_Id"], 0.0)
            rows[r["Kernel_Name"].split("(")[0]][r["Dispatch_Id"]] += float(r["Counter_Value"])
known = {"k_gather64": 2048.0, "k_gather64_half": 1024.0, "k_stream": 2048.0}  # MiB actually read
res = {}
for k, per in rows.items():
    v = sorted(per.values())
    name = k.replace("void ", "")
    if name in known:
        mib = v[-1] / 1024.0   # FETCH_SIZE is in KiB
        res[name] = {"fetch_size_mib": mib, "known_mib": known[name], "ratio_reported_over_known": mib / known[name]}
        print(f"{name:18s} FETCH_SIZE {mib:9.1f} MiB   known {known[name]:7.1f} MiB   reported/known = {mib / known[name]:.3f}")
json.dump(res, open(out + "/calib_fetch.json", "w"), indent=1)
PY
rm -f "$OUT/calib_fetch"
find "$OUT" -type f -size +5M -delete
