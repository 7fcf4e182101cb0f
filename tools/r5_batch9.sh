#!/bin/bash
mkdir -p gpurun_out
step() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > gpurun_out/$log 2>&1; local rc=$?; echo "[$log] rc=$rc"; tail -3 gpurun_out/$log; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT: batch ends"; exit 1; fi; }
export N64=$PWD/build_variants/libhrcore_n64.so
step 800 r5m_gpu_suite.log python -m pytest tests -m gpu -x -q
for k in 20 128; do
  timeout -k 10 400 python bench.py --steps $k --warmup 5 --no-converge --cpu-seconds 0 --parity-seconds 0 > gpurun_out/r5m_counters_node32_$k.json 2> gpurun_out/r5m_err.txt || tail -3 gpurun_out/r5m_err.txt
  HRCORE_LIB=$N64 timeout -k 10 400 python bench.py --steps $k --warmup 5 --no-converge --cpu-seconds 0 --parity-seconds 0 > gpurun_out/r5m_counters_node64_$k.json 2> gpurun_out/r5m_err.txt || tail -3 gpurun_out/r5m_err.txt
done
python - <<'PY'
import json
for k in (20, 128):
    for v in ("node64", "node32"):
        try:
            d = json.load(open(f"gpurun_out/r5m_counters_{v}_{k}.json"))
        except Exception as e:
            print(v, k, "failed", e); continue
        r = d["roofline"]; u = r.get("units") or {}
        kt = (u.get("per_kernel") or {}).get("k_trace", {})
        print(f"c3 {k} passes {v}: {d['value']:.1f} Mrays/s, k_trace avg launch {r.get('avg_launch_ms_device_clock', 0):.3f} ms, TA busy {kt.get('ta_busy')}, VALU busy {kt.get('valu_busy')}, "
              f"TA wave loads/ray {u.get('ta_wave_loads_per_ray')}, VALU instr/ray {(r.get('valu') or {}).get('wave_instructions_per_ray')}, frac_k_trace {r.get('frac_k_trace')}, hbm bytes/ray {r.get('hbm_side_bytes_per_ray')}")
PY
run() { local label=$1 wl=$2 k=$3 lib=$4
  for i in 1 2 3; do
    v=$(HRCORE_LIB=$lib timeout -k 10 300 python bench.py --quick --parity-seconds 0 --workload $wl --steps $k --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")
    echo "[$label] $wl $k passes: $v" >> gpurun_out/r5m_node32_workloads.txt
  done
}
for wl in c3d terrain c2 c5; do for k in 20 128; do
  run node64 $wl $k $N64
  run node32 $wl $k ""
done; done
cat gpurun_out/r5m_node32_workloads.txt
