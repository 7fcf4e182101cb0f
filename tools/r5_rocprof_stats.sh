#!/bin/bash
# ON THE GPU BOX: rocprofv3 --kernel-trace --stats of the bench command with no wake-up and no warm-up, so that every k_trace dispatch
# rocprofv3 sees belongs to the timed region; the same run's HIP-event average is in the JSON next to it.  HR_TUNE=packets=1,corun=2 (terrain: corun=0): what the packet
# selector chooses for these four workloads once its first probe has reported (a run without warm-up would spend its first batches before that).
ROOT="$PWD"; cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
python3 bench.py --quick --steps 16 > /dev/null 2>&1
for cfg in "r5z:--steps 20" "r5z128:--steps 128" "r5_c3d:--workload c3d --steps 20" "r5_terrain:--workload terrain --steps 20"; do
  tag="${cfg%%:*}"; args="${cfg#*:}"
  tune="packets=1,corun=2"; [ "$tag" = r5_terrain ] && tune="packets=1,corun=0" # (beside k_trace where the selector's probe puts it there: c3, c3d)
  HR_TUNE=$tune HR_BENCH_TIME_KERNELS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "gpurun_out/prof_$tag" -o t --output-format csv -- python3 bench.py --quick --no-wakeup --warmup 0 $args > "gpurun_out/${tag}_bench_under_rocprof.json" 2> "gpurun_out/${tag}_rocprof.err" || { echo "rocprof $tag failed"; tail -3 "gpurun_out/${tag}_rocprof.err"; }
  s=$(find "gpurun_out/prof_$tag" -name '*kernel_stats.csv' | head -1); [ -n "$s" ] && cp "$s" "gpurun_out/${tag}_kernel_stats.csv"
  rm -rf "gpurun_out/prof_$tag"
  python3 - "$tag" <<'PY'
import csv, json, sys
tag = sys.argv[1]
d = json.loads([l for l in open(f"gpurun_out/{tag}_bench_under_rocprof.json") if l.startswith("{")][-1])
k, n = d["extra"]["kernel_ms_rank0"], d["extra"]["kernel_launches_rank0"]
rows = list(csv.DictReader(open(f"gpurun_out/{tag}_kernel_stats.csv")))
row = [r for r in rows if "k_trace<" in r["Name"]][0]
print(f"{tag}: rocprofv3 k_trace {row['Calls']} calls, average {float(row['AverageNs'])/1e6:.4f} ms | HIP events of the same run: {n['trace']} launches, average {k['trace']/n['trace']:.4f} ms | "
      f"device clock {d['roofline']['avg_launch_ms_device_clock'] if d['roofline'] else float('nan'):.4f} ms | {d['value']:.1f} Mrays/s under the profiler")
pk = [r for r in rows if "k_raygen_packets" in r["Name"]]
if pk:  # (with the packet kernel on its second stream the event bucket "raygen" is empty: the kernel is not bracketed there)
    print(f"{tag}:   k_raygen_packets (camera rays: generation + traversal) {pk[0]['Calls']} calls, average {float(pk[0]['AverageNs'])/1e6:.4f} ms; camera rays {d['extra']['camera_rays']}")
PY
done
