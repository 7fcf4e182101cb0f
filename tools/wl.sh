#!/bin/bash
# tools/wl.sh "<workloads>" "<step counts>" [extra bench args]  — one line per (workload, steps): Mrays/s, ms/step, launches
for w in $1; do for st in $2; do
  python bench.py --quick --workload $w --steps $st --warmup 5 $3 > gpurun_out/wl.json 2> gpurun_out/wl.err || { echo "FAILED $w $st"; tail -3 gpurun_out/wl.err; continue; }
  python - "$w" "$st" <<'PY'
import json, sys
d = json.load(open("gpurun_out/wl.json"))
k, n = d["extra"]["kernel_ms_rank0"], d["extra"]["kernel_launches_rank0"]
print(f"{sys.argv[1]:5s} steps {sys.argv[2]:>4s}  {d['value']:8.1f} Mrays/s  {d['ms_per_step']:.3f} ms/step  trace {k['trace']:.1f} ms x{n['trace']}  shade {k['shade']:.1f}  raygen {k['raygen']:.1f}  resolve {k['resolve']:.1f}")
PY
done; done
