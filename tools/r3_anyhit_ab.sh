#!/bin/bash
# Runs ON THE GPU BOX: unordered node steps for occlusion rays (default build) against ordered ones (build_variants/libhrcore_anyord.so):
# throughput, traversal counters, and VALU instructions of k_trace (rocprofv3 --pmc SQ_INSTS_VALU) per ray.
ROOT="$PWD"
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
python3 bench.py --quick --steps 16 > /dev/null 2>&1 || true
for v in unordered ordered; do
  lib=""; [ $v = ordered ] && lib="$ROOT/build_variants/libhrcore_anyord.so"
  for rep in 1 2; do
    HRCORE_LIB="$lib" python3 bench.py --quick --steps 128 > gpurun_out/anyab.json 2>/dev/null
    python3 - $v <<'PY'
import json, sys
d = json.load(open("gpurun_out/anyab.json")); k = d["extra"]["kernel_ms_rank0"]; n = d["extra"]["kernel_launches_rank0"]
print(f"{sys.argv[1]:10s} 128 steps: {d['value']:8.1f} Mrays/s  trace {k['trace']/n['trace']:.3f} ms x{n['trace']}")
PY
  done
  HRCORE_LIB="$lib" python3 bench.py --cpu-seconds 0 --no-pmc --no-converge --steps 20 --warmup 5 > gpurun_out/anyab_s.json 2>/dev/null
  d="gpurun_out/anyab_$v"
  HRCORE_LIB="$lib" timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d "$d" -o p --output-format csv -- python3 bench.py --quick --steps 20 --warmup 0 --no-wakeup > "$d.json" 2> "$d.err" || { echo "pmc $v failed"; tail -3 "$d.err"; }
  python3 - $v "$d" "$d.json" gpurun_out/anyab_s.json <<'PY'
import csv, glob, json, os, sys
v, d, j, sj = sys.argv[1:5]
tot = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_trace" in row["Kernel_Name"]:
            tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
b = json.load(open(j)); s = json.load(open(sj))
rays = b["extra"]["rays"]; closest = b["extra"]["closest_rays"]; occl = rays - closest
c = s["extra"]["gpu_traversal_counters"]
print(f"{v:10s} k_trace SQ_INSTS_VALU {tot.get('SQ_INSTS_VALU', 0):.4e} over {rays:.0f} rays ({occl:.0f} occlusion) = {tot.get('SQ_INSTS_VALU', 0) / rays:.1f} wave-instructions per ray;"
      f" node visits per closest ray {c['node4_visits_per_closest_ray']:.2f}, per occlusion ray {c['node4_visits_per_occlusion_ray']:.2f}; driver-style {s['value']:.1f} Mrays/s")
print(f"           VALU_TOTAL {tot.get('SQ_INSTS_VALU', 0):.6e} RAYS {rays:.0f} OCCL {occl:.0f} VC {c['node4_visits_per_closest_ray']:.4f} VA {c['node4_visits_per_occlusion_ray']:.4f}")
PY
  rm -rf "$d"
done
