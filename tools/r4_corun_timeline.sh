#!/bin/bash
# ON THE GPU BOX: do the packet kernel and k_trace of a step really overlap?  rocprofv3 --kernel-trace of the default 128-pass run; per step:
# start and end of both kernels on the device's clock and the share of the packet kernel's run time during which its step's k_trace runs too.
ROOT="$PWD"; cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
rm -rf gpurun_out/tl && timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/tl -o t --output-format csv -- python3 bench.py --quick --steps 128 > gpurun_out/tl_bench.json 2> gpurun_out/tl.err || { tail -3 gpurun_out/tl.err; exit 1; }
python3 - <<'PY'
import csv, glob, json
f = glob.glob("gpurun_out/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f))]
pk = [(s, e) for n, s, e in rows if "k_raygen_packets" in n]
tr = [(s, e) for n, s, e in rows if "k_trace<" in n]
d = json.loads([l for l in open("gpurun_out/tl_bench.json") if l.startswith("{")][-1])
print(f"bench under the tracer: {d['value']:.1f} Mrays/s, {d['extra']['camera_rays']}")
t0 = min(s for s, _ in pk + tr)
tot = ov = 0
for k, (s, e) in enumerate(pk[-10:]):
    o = sum(max(0, min(e, te) - max(s, ts)) for ts, te in tr)
    mate = max(tr, key=lambda x: max(0, min(e, x[1]) - max(s, x[0])))
    tot += e - s; ov += o
    print(f"packet kernel {(s-t0)/1e6:9.3f} .. {(e-t0)/1e6:9.3f} ms ({(e-s)/1e6:6.3f} ms)   k_trace beside it {(mate[0]-t0)/1e6:9.3f} .. {(mate[1]-t0)/1e6:9.3f} ms ({(mate[1]-mate[0])/1e6:6.3f} ms)   overlap {o/(e-s)*100:5.1f} % of the packet kernel")
print(f"last {min(10, len(pk))} packet launches: {ov/tot*100:.1f} % of their run time is beside a k_trace launch; {len(pk)} packet launches, {len(tr)} k_trace launches in the process")
PY
rm -rf gpurun_out/tl
