#!/bin/bash
# usage: tools/ab.sh "<bench args>" "<HR_TUNE 1>" "<HR_TUNE 2>" ...   — one compact line per variant
args="$1"; shift
# a variant "@NAME:tune" runs build_variants/libhrcore_NAME.so instead of the in-tree library
for t in "$@"; do
  lib=""
  if [[ "$t" == @* ]]; then n="${t%%:*}"; lib="$PWD/build_variants/libhrcore_${n#@}.so"; fi
  tune="${t#@*:}"; [[ "$t" == @* && "$t" != *:* ]] && tune=""
  HRCORE_LIB="$lib" HR_TUNE="$tune" python bench.py --quick $args > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "FAILED $t"; tail -3 gpurun_out/ab.err; exit 1; }
  python - "$t" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab.json"))
k = d["extra"]["kernel_ms_rank0"]; n = d["extra"]["kernel_launches_rank0"]
print(f"{sys.argv[1]:32s} {d['value']:8.1f} Mrays/s  {d['ms_per_step']:.3f} ms/step  trace {k['trace']/max(n['trace'],1):.3f} ms x{n['trace']}  shade {k['shade']/max(n['shade'],1):.3f}  raygen {k['raygen']/max(n['raygen'],1):.3f}")
PY
done
