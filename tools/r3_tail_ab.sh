#!/bin/bash
# Runs ON THE GPU BOX: per-launch k_trace / k_shade durations (rocprofv3 --kernel-trace) of a 20-step run of a 1/8 shard
# for several HR_TUNE variants.   tools/r3_tail_ab.sh <tag> "<bench extra args>" "<tune 1>" "<tune 2>" ...
tag="$1"; shift; extra="$1"; shift
ROOT="$PWD"
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
python3 bench.py --quick --steps 16 > /dev/null 2>&1 || true
i=0
for t in "$@"; do
  i=$((i+1))
  lib=""
  if [[ "$t" == @* ]]; then n="${t%%:*}"; lib="$PWD/build_variants/libhrcore_${n#@}.so"; fi
  tune="${t#@*:}"; [[ "$t" == @* && "$t" != *:* ]] && tune=""
  d="gpurun_out/${tag}_v$i"
  HRCORE_LIB="$lib" HR_TUNE="$tune" timeout -k 10 200 rocprofv3 --kernel-trace -d "$d" -o t --output-format csv -- python3 bench.py --quick --warmup 5 --no-wakeup --steps 20 $extra > "$d.json" 2> "$d.err" || { echo "variant $t failed"; tail -5 "$d.err"; exit 1; }
  f=$(find "$d" -name '*kernel_trace.csv' | head -1)
  python3 - "$t" "$f" "$d.json" <<'PY'
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[2])) if any(k in r["Kernel_Name"] for k in ("k_trace", "k_shade", "k_raygen", "k_tail"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed region starts at the last-but-... raygen group: print every launch after the warm-up's final resolve
d = json.load(open(sys.argv[3]))
names = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hr::", "").split("<")[0] for r in rows]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
n_tr = d["extra"]["kernel_launches_rank0"]["trace"]
idx = [i for i, n in enumerate(names) if n == "k_trace"][-n_tr:]
first = idx[0]
span = (int(rows[-1]["End_Timestamp"]) - int(rows[first]["Start_Timestamp"])) / 1e3
print(f"{sys.argv[1]:28s} {d['value']:8.1f} Mrays/s {d['ms_per_step']:.4f} ms/step | span {span:8.1f} us")
print("   trace:", " ".join(f"{dur[i]:.0f}" for i in range(first, len(rows)) if names[i] == "k_trace"))
print("   shade:", " ".join(f"{dur[i]:.0f}" for i in range(first, len(rows)) if names[i] == "k_shade"))
other = [f"{names[i]}:{dur[i]:.0f}" for i in range(first, len(rows)) if names[i] not in ("k_trace", "k_shade")]
if other: print("   other:", " ".join(other))
PY
  rm -rf "$d"
done
