#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the per-pass ray kernel at 1920x1080, 8 bounces (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W           (N > 1: launched by torch.distributed.run)

A "step" is one pass (one sample per pixel) of the hot path — primary-ray generation, BVH traversal,
PBR/glass shading with NEE + environment, accumulation — over the whole frame; the scene, BVH, sample
tables and the accumulation buffer are resident in HBM before the timed region.  With N > 1 the frame is
sharded by 32x32-pixel tile across the ranks (strong scaling: the frame is fixed); every step each rank's
owned tiles are gathered on rank 0 over RCCL on a side stream, and the assembled final image is checked
inside the timed region.  A small shard batches several passes into each kernel launch.

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel (k_trace: closest-hit + occlusion traversal).
Its fraction is COUNTER-BASED and cannot exceed 1: before the parent process touches the GPU, the same command
is run under `rocprofv3 --pmc` in child processes (FETCH_SIZE, WRITE_SIZE and SQ_INSTS_VALU in separate passes, as
MI355X_MICROARCH.md prescribes), which yields HBM-side bytes and VALU instructions per ray of THIS configuration and
step count; multiplied by the rays of the timed region and divided by its wall time they give the achieved bandwidth
and issue rate against the 8 TB/s and 256 CU x 4 SIMD x f / 2 ceilings.  The algorithmic (spec-BVH) figure of SURVEY
§8d is kept under `spec_model`.  `cpu_baseline` is the CPU oracle (oracle/, the checker — used here only as the
reported baseline leg, built -O3 -march=native) timed on a bounded tile sample of the same workload.
"""
import argparse
import csv
import glob
import json
import math
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from heatray_amd import _ffi as ffi  # noqa: E402
from heatray_amd import core, scenes, tiles  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
# VALU issue ceiling: a wave64 VALU instruction occupies a SIMD-32 for 2 cycles once more than one wave is resident
# (MI355X_MICROARCH.md 'Wave scheduling'; tools/calib_valu.hip measures it: profiles/r2_calib_valu.json), 4 SIMDs per CU
VALU_CLOCK_GHZ = 2.4
VALU_CYCLES_PER_WAVE_OP = 2.0

WORKLOADS = {
    # BASELINE.json configs; c3 is the configuration the north-star target (>= 1 Gray/s) is quoted on
    "c1": dict(desc="Cornell box (32 tris), 256x256, 4 bounces"),
    "c2": dict(desc="synthetic soup 50k tris, 16 PBR materials, 1 directional light, 1920x1080, 8 bounces"),
    "c2p": dict(desc="c2 with 30% single-sided / alpha-masked (pass-through) materials, as glTF assets have them"),
    "c3": dict(desc="synthetic soup 1M tris + 2048x1024 HDRI + NEE, 1920x1080, 8 bounces"),
    # the regime the metric's name promises — paths that really bounce: the c3 soup and lights inside a closed room with a skylight
    "c3d": dict(desc="c3's soup (1M tris) + HDRI + sun inside a closed diffuse room [-2,2]^3 with a 1.2x1.2 skylight, camera inside, 1920x1080, 8 bounces"),
    "c5": dict(desc="soup 1M tris, 25% glass, 25% clearcoat, f/2.8 pentagon-bokeh DoF, 3840x2160, 16 bounces"),
    # SURVEY 8d's coherent counterpart of the soup: an indexed, shared-vertex grid mesh
    "terrain": dict(desc="indexed terrain grid mesh 1000x500 quads (1M tris, shared vertices) + 2048x1024 HDRI + NEE, 1920x1080, 8 bounces"),
}


def build_scene(name, width, height, passes):
    if name == "c1":
        return scenes.cornell_box(width or 256, height or 256, bounces=4, passes=passes)
    if name == "c2":
        return scenes.triangle_soup(50_000, width or 1920, height or 1080, bounces=8, passes=passes, env=False)
    if name == "c2p":
        return scenes.triangle_soup(50_000, width or 1920, height or 1080, bounces=8, passes=passes, env=False, passthrough_fraction=0.3)
    if name == "c3":
        return scenes.triangle_soup(1_000_000, width or 1920, height or 1080, bounces=8, passes=passes, env=True)
    if name == "c3d":
        return scenes.triangle_soup(1_000_000, width or 1920, height or 1080, bounces=8, passes=passes, env=True, room=True)
    if name == "terrain":
        return scenes.terrain(1000, 500, width or 1920, height or 1080, bounces=8, passes=passes, env=True)
    if name == "c5":
        sc = scenes.triangle_soup(1_000_000, width or 3840, height or 2160, bounces=16, passes=passes, env=True,
                                  glass_fraction=0.25, clearcoat_fraction=0.25)
        sc.options.fstop = 2.8
        sc.options.bokeh_shape = ffi.HR_BOKEH_PENTAGON  # util::randomPolygonal(5 edges), generated on the device (hr_sequences_generate)
        return sc
    raise SystemExit(f"unknown workload {name}")


def cpu_baseline(sc, budget_s, lut):
    """The CPU oracle (kind "port") on the host cores, on a bounded sample of the same workload: an interleaved
    1/world shard of the frame's 32x32 tiles (sized from a one-second probe), rendered pass after pass until
    `budget_s` seconds of CPU work are done."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    n_tiles = ((sc.width + 31) // 32) * ((sc.height + 31) // 32)
    cores = oracle_lib.usable_cpus()  # affinity mask capped by the cgroup CPU quota (a one-GPU box: 16 of the host's 256)
    max_passes = sc.options.max_render_passes

    def run(world, budget, threads=0):
        eng = oracle_lib.engine(threads=threads or cores, native=True, rank=0, world=world, tile_size=32)
        t0 = time.perf_counter()
        sc.apply(eng, lut=lut)
        build_s = time.perf_counter() - t0
        eng.render_pass(sc.options.pass_params(0))  # untimed: first touch of the BVH
        eng.clear()
        passes, t0 = 0, time.perf_counter()
        while True:
            eng.render_pass(sc.options.pass_params(passes % max_passes))  # sample indices stay inside the tables
            passes += 1
            el = time.perf_counter() - t0
            if el > budget:
                break
        return eng.stats(), passes, el, build_s

    world = max(1, min(64, n_tiles))
    st, passes, el, build_s = run(world, 1.0)  # probe
    per_pass_full = el / passes * world
    # shard so that one pass takes at most a quarter of the budget (at least 1/64 of the frame, at most all of it),
    # then render passes until the budget is used up
    world = int(max(1, min(64, math.ceil(per_pass_full / max(budget_s / 4.0, 1e-3)))))
    st, passes, el, build_s = run(world, budget_s)
    rays = st.rays_closest + st.rays_any
    # one thread on a 1/64 shard for two seconds: the per-core figure BASELINE.md asks for (and a determinism check: the
    # oracle's result does not depend on its thread count, tests/test_oracle_render.py)
    st1, passes1, el1, _ = run(64, min(2.0, budget_s), threads=1)
    return {
        "single_thread_value": (st1.rays_closest + st1.rays_any) / el1 / 1e6,
        "value": rays / el / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
        "build": "oracle/Makefile native: g++ -O3 -march=native -fopenmp (timing build; the parity build keeps -O2 -ffp-contract=off)",
        "sample": f"1/{world} of the frame's 32x32 tiles (interleaved), {passes} passes, {rays} rays in {el:.1f} s "
                  f"(+{build_s:.1f} s scene/BVH build), OpenMP over tiles on {cores} threads",
        "per_core": rays / el / 1e6 / cores,
        "V": (st.node_visits - st.node_visits_any) / max(st.rays_closest, 1),
        "T": (st.tri_tests - st.tri_tests_any) / max(st.rays_closest, 1),
        "V_any": st.node_visits_any / max(st.rays_any, 1), "T_any": st.tri_tests_any / max(st.rays_any, 1),
    }


def parity_leg(sc, eng_frame, first_pass, n_passes, budget_s, world_label):
    """GPU vs the CPU oracle (the PARITY build: -O2 -ffp-contract=off, the checker of tests/) on the frame the timed region itself
    produced: the oracle renders the same passes of the same scene and the HDR buffers are compared pixel for pixel.  The whole
    frame when that fits `budget_s` of host time (a one-pass probe decides); otherwise an interleaved 1/k shard of the frame's
    32x32 tiles, all passes.  BASELINE.md §2: "rel-L2 of GPU HDR buffer vs oracle" per configuration."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    t0 = time.perf_counter()
    o = oracle_lib.engine()
    sc.apply(o)  # tables and LUT from the oracle's own generators (bit-identical to the device's: tests/test_gpu_parity.py)
    t1 = time.perf_counter()
    o.render_pass(sc.options.pass_params(first_pass))
    per_pass = time.perf_counter() - t1
    k = 1
    if per_pass * n_passes > budget_s:
        k = int(min(64, math.ceil(per_pass * n_passes / budget_s)))
        o.close()
        o = oracle_lib.engine(rank=k // 2, world=k, tile_size=32)
        sc.apply(o)
        o.render_pass(sc.options.pass_params(first_pass))
    for i in range(1, n_passes):
        o.render_pass(sc.options.pass_params(first_pass + i))
    ref = o.readback()
    o.close()
    own = np.ones(ref.shape[:2], dtype=bool) if k == 1 else (tiles.owner_map(sc.width, sc.height, k) == k // 2)
    g, r = eng_frame[own].astype(np.float64), ref[own].astype(np.float64)
    differing = int((eng_frame[own] != ref[own]).any(axis=-1).sum())
    return {"rel_l2": float(np.linalg.norm(g - r) / max(np.linalg.norm(r), 1e-30)), "bit_exact": differing == 0, "pixels": int(own.sum()),
            "differing_pixels": differing, "passes": n_passes, "tolerance_rel_l2": 1e-4,
            "frame": f"the timed region's own output{world_label}: passes {first_pass}..{first_pass + n_passes - 1}, "
                     + ("every pixel of the frame" if k == 1 else f"the interleaved tile shard {k // 2} of {k} (32x32 tiles over the whole frame)"),
            "checker": "oracle/liboracle.so (parity build), OpenMP on the host cores", "seconds": time.perf_counter() - t0}


def _short_kernel(name):
    n = name.split("(")[0].replace("void ", "").replace("hr::", "")
    return n.split("<")[0]


PMC_PASSES = (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]),
              # VALU wave-instructions by issue class (the SQ block has eight slots per pass): plain f32 add / mul / fma issue at the double rate,
              # transcendentals at a quarter, everything else (conversions, min / max, compares, integer and bit-field work) at the single rate
              ("sq", ["SQ_INSTS_VALU", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_THREAD_CYCLES_VALU"]),
              # busy cycles of the two units that bind k_trace: the texture addresser of a CU, and the VALUs
              ("units", ["TA_TA_BUSY_sum", "GRBM_GUI_ACTIVE", "SQ_ACTIVE_INST_VALU", "TA_FLAT_READ_WAVEFRONTS_sum"]))


def pmc_legs(args, keep_dir=None):
    """Counter passes of THIS command (same workload, size, depth and step count), each in a child process under
    `rocprofv3 --kernel-trace --pmc <counters>` — run before the parent initialises the GPU.  FETCH_SIZE and WRITE_SIZE do
    not fit one pass (MI355X_MICROARCH.md 'rocprofv3 PMC slots'), so they are separate passes; the SQ counters are a third.
    Returns per-kernel counter sums over every dispatch of the child's run plus the child's ray count, or None when rocprofv3
    is not usable here (then the roofline falls back to the committed per-ray figures and says so)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    out_root = os.path.abspath(keep_dir) if keep_dir else tempfile.mkdtemp(prefix="hr_pmc_", dir="/tmp")
    os.makedirs(out_root, exist_ok=True)
    # The children run the SAME configuration as the timed region — wake-up, W warm-up steps, K counted steps — and only the dispatches
    # of their timed region are summed (the last `launches` k_trace dispatches and everything launched from the first of them on):
    # busy fractions and bytes per ray then describe a warm device at the step count the line is quoted on, not a cold process's
    # first milliseconds (on a fresh box those run up to a fifth slower).
    child = [sys.executable if os.path.basename(sys.executable).startswith("python") else "python3", os.path.join(ROOT, "bench.py"),
             "--pmc-child", "--workload", args.workload, "--steps", str(args.steps), "--warmup", str(args.warmup), "--cpu-seconds", "0", "--parity-seconds", "0", "--no-stats-pass",
             "--no-pmc", "--no-converge", "--estimator", args.estimator, "--width", str(args.width), "--height", str(args.height), "--depth", str(args.depth)]
    env = dict(os.environ, TMPDIR="/tmp")
    res = {"kernels": {}, "passes": {}, "dir": out_root if keep_dir else None}
    t0 = time.perf_counter()
    for name, ctrs in PMC_PASSES:
        d = os.path.join(out_root, name)
        shutil.rmtree(d, ignore_errors=True)
        cmd = [exe, "--kernel-trace", "--pmc", *ctrs, "-d", d, "-o", name, "--output-format", "csv", "--", *child]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=200)
        except (OSError, subprocess.TimeoutExpired) as e:
            res["passes"][name] = {"error": str(e)[:200]}
            continue
        line = [l for l in r.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            res["passes"][name] = {"error": f"rc {r.returncode}: " + r.stderr.decode(errors="replace")[-300:]}
            continue
        cj = json.loads(line[-1])
        n_trace = int(cj["extra"]["kernel_launches_rank0"]["trace"])
        rows = []
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    rows.append((int(row["Dispatch_Id"]), _short_kernel(row["Kernel_Name"]), row["Counter_Name"], float(row["Counter_Value"])))
        trace_ids = sorted({r_[0] for r_ in rows if r_[1] == "k_trace"})
        first = trace_ids[-n_trace] if 0 < n_trace <= len(trace_ids) else (trace_ids[0] if trace_ids else 0)
        # the timed region starts with the ray generation of its first macro step, launched right before that k_trace
        gens = [r_[0] for r_ in rows if r_[1] in ("k_raygen", "k_raygen_packets") and r_[0] < first and r_[0] > (trace_ids[-n_trace - 1] if n_trace < len(trace_ids) else -1)]
        start = min(gens) if gens else first
        res["passes"][name] = {"rays": cj["extra"]["rays"], "steps": cj["steps"], "warmup": cj["warmup"], "counters": ctrs, "k_trace_launches_counted": n_trace,
                               "dispatches_in_process": len({r_[0] for r_ in rows}), "dispatches_counted": len({r_[0] for r_ in rows if r_[0] >= start})}
        # the same dispatches' durations by the profiler's timestamps (kernel trace of the same pass): what the counters of this pass are
        # normalised with where a busy FRACTION is wanted (cycle counters of another pass would bring that pass's clock with them)
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    try:
                        did = int(row["Dispatch_Id"])
                    except (KeyError, ValueError):
                        continue
                    if did >= start:
                        kk = res["kernels"].setdefault(_short_kernel(row["Kernel_Name"]), {})
                        kk["duration_ns_" + name] = kk.get("duration_ns_" + name, 0.0) + float(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        for disp_id, k, ctr, val in rows:
            if disp_id < start:
                continue
            kk = res["kernels"].setdefault(k, {})
            kk[ctr] = kk.get(ctr, 0.0) + val
            kk.setdefault("_dispatches_" + ctr, set()).add(disp_id)
        if not keep_dir:
            shutil.rmtree(d, ignore_errors=True)
    for k in res["kernels"].values():
        for key in [x for x in k if x.startswith("_dispatches_")]:
            k["launches_" + key[len("_dispatches_"):]] = len(k.pop(key))
    res["seconds"] = time.perf_counter() - t0
    if not keep_dir:
        shutil.rmtree(out_root, ignore_errors=True)
    ok = all("rays" in res["passes"].get(n, {}) for n, _ in PMC_PASSES[:2])
    return res if ok else {"failed": True, **res}


def convergence_leg(core, sc, dev, stream, cap, n_runs, budget_s=None):
    """BASELINE metric 2 (heatray_amd/convergence.py): err(n) is read off the accumulation buffer after every render_pass
    WITHOUT draining the pipeline — the buffer always holds complete passes, in order, and its alpha says how many.

    Two engines share the work.  The library collects 12 passes of a 1080p frame per pipeline step, so the buffer of an engine with the
    default scheduling advances 12 passes at a time: that engine (`fast`) renders the reference image and, in every run, finds the
    12-pass window in which err crosses the threshold; a second engine that injects one pass per step (`HR_TUNE batch=1`, 25 % slower per
    pass) then continues from a copy of the buffer at the window's start, completing each pass before the buffer is read, and gives err(n) at
    every n inside it.  The image does not depend
    on the scheduling (tests/test_gpu_parity.py::test_pipeline_scheduling_is_result_invariant), so the numbers are those of a run at
    one-pass resolution throughout — which is what this leg did before, in 40 % more time."""
    import torch
    from heatray_amd import convergence as cv
    passes_before = sc.options.max_render_passes
    sc.options.max_render_passes = cv.REFERENCE_PASSES          # sample tables long enough for the reference render
    fast = core.create_engine(device_id=dev.index, stream=stream)
    tune_before = os.environ.get("HR_TUNE")
    os.environ["HR_TUNE"] = (tune_before + "," if tune_before else "") + "batch=1"
    try:
        fine = core.create_engine(device_id=dev.index, stream=stream)
    finally:
        if tune_before is None:
            del os.environ["HR_TUNE"]
        else:
            os.environ["HR_TUNE"] = tune_before
    sc.apply(fast)
    sc.apply(fine)
    fb = torch.zeros((sc.height, sc.width, 4), dtype=torch.float32, device=dev)
    fb_fine = torch.zeros_like(fb)
    fast.bind_external_frame(fb.data_ptr())
    fine.bind_external_frame(fb_fine.data_ptr())
    t0 = time.perf_counter()
    for n in range(cv.REFERENCE_PASSES):
        fast.render_pass(sc.options.pass_params(n))
    fast.flush()
    torch.cuda.synchronize()
    ref = cv.normalised(fb).clone()
    ref_d = ref.double()
    ref_norm = ref_d.norm()
    t_ref = time.perf_counter() - t0

    def err_of(buf):
        return (cv.normalised(buf).double() - ref_d).norm() / ref_norm

    results, curves, t0 = [], [], time.perf_counter()
    chunk = 64
    for run in range(n_runs):
        if budget_s is not None and run >= 3 and time.perf_counter() - t0 + t_ref > budget_s:
            break  # (out of time: the statistic is then over the runs made, and the line says how many)
        offsets = cv.offsets_table(fast, run, sc.width, sc.height)
        fast.set_seq_offsets(offsets)
        fine.set_seq_offsets(offsets)
        fast.clear()
        torch.cuda.synchronize()
        window, issued, curve, next_pow = None, 0, {}, 1
        keep_snap, keep_count = torch.zeros_like(fb), 0             # the buffer at the newest boundary of the chunks before
        while window is None and issued < cap:
            errs = torch.full((chunk,), float("inf"), dtype=torch.float64, device=dev)
            counts = torch.zeros((chunk,), dtype=torch.float32, device=dev)
            snaps = []
            for k in range(chunk):
                fast.render_pass(sc.options.pass_params(issued))
                issued += 1
                snaps.append(fb.clone())
                errs[k] = err_of(snaps[-1])
                counts[k] = snaps[-1][0, 0, 3]                       # complete passes in the buffer (same for every pixel)
            e, c = errs.cpu().numpy(), [int(x) for x in counts.cpu().numpy()]
            for er, cn in zip(e, c):
                if cn >= next_pow:                                   # err(n) at the first 12-pass boundary at or after each power of two, for the report
                    curve[cn] = float(er)
                    while next_pow <= cn:
                        next_pow *= 2
            hit = [cn for er, cn in zip(e, c) if cn > 0 and er <= cv.THRESHOLD]
            if hit:
                upper = min(hit)                                     # first boundary at or below the threshold
                below = [(cn, k) for k, cn in enumerate(c) if cn < upper]
                lower, snap = (max(below)[0], snaps[max(below)[1]]) if below and max(below)[0] >= keep_count else (keep_count, keep_snap)
                window = (lower, upper, snap)
            else:
                keep_snap, keep_count = snaps[-1], c[-1]
            del snaps
        found = None
        if window is not None:
            lower, upper, snap = window
            # from the window's start, one pass at a time: each is completed (flush) before the buffer is read, so every n is seen
            fb_fine.copy_(snap)
            extra = upper - lower
            errs = torch.full((extra,), float("inf"), dtype=torch.float64, device=dev)
            counts = torch.zeros((extra,), dtype=torch.float32, device=dev)
            for k in range(extra):
                fine.render_pass(sc.options.pass_params(lower + k))
                fine.flush()
                errs[k] = err_of(fb_fine)
                counts[k] = fb_fine[0, 0, 3]
            e, c = errs.cpu().numpy(), [int(x) for x in counts.cpu().numpy()]
            hit = [cn for er, cn in zip(e, c) if cn > lower and er <= cv.THRESHOLD]
            found = min(hit) if hit else upper
        results.append(found)
        curves.append(curve)
    t_runs = time.perf_counter() - t0
    sc.options.max_render_passes = passes_before
    fast.close()
    fine.close()
    ns = sorted(set().union(*[set(c) for c in curves]))
    med = {str(n): float(np.median([c[n] for c in curves if n in c])) for n in ns}
    converged = [r for r in results if r is not None]
    return {"p50": cv.p50(results, cap) if len(converged) * 2 > len(results) else None, "threshold_rel_l2": cv.THRESHOLD, "runs": results, "cap": cap,
            "runs_made": len(results), "runs_asked": n_runs,
            "median_err_at_pass": med,
            "reference_passes": cv.REFERENCE_PASSES, "reference_render_s": t_ref, "runs_s": t_runs,
            "note": "p50 is null when fewer than half of the runs reached the threshold within the cap", "definition": "min n with ||I_n - I_ref|| / ||I_ref|| <= 0.02 on RGB/A; 16 runs = Sobol sequence index 0..15 of the "
                          "SequenceOffsets table; I_ref = 8192 passes with index 0 (SURVEY 8d metric 2); err(n) at every n inside the 12-pass window "
                          "in which it crosses the threshold"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128, help="timed passes (SURVEY 8d: at least 32); the pass pipeline is depth+2 stages deep, so short runs weigh its fill and drain")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="time budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--shard-of", type=int, default=0, help="tuning aid, single process: render only rank 0's tiles of a W-way "
                    "sharded frame (no collective); the JSON line is marked emulated and is not a benchmark result")
    ap.add_argument("--converge", action="store_true", help="passes-to-converge p50 (BASELINE metric 2): no time budget for the leg (all --converge-runs "
                    "runs whatever they take); N = 1 only")
    ap.add_argument("--converge-budget", type=float, default=150.0, help="wall-time budget in seconds of the convergence leg of the default run (reference "
                    "render + runs; the leg stops early after >= 3 runs when it is used up and reports the number of runs made)")
    ap.add_argument("--estimator", default="reference", choices=["reference", "env_mis", "all_lights"], help="estimator of the timed region and of the convergence "
                    "leg: the reference's (BASELINE metric), or importance-sampled environment + one-sample MIS (include/hrcore.h)")
    ap.add_argument("--no-converge", action="store_true", help="skip the live passes-to-converge leg (the committed measurement is quoted, marked as such)")
    ap.add_argument("--converge-mis", action="store_true", help="also run the convergence leg with the env-MIS estimator")
    ap.add_argument("--converge-cap", type=int, default=4096, help="give up a convergence run after this many passes")
    ap.add_argument("--converge-runs", type=int, default=16, help="number of runs (Sobol sequence indices 0..n-1 of the offsets table)")
    ap.add_argument("--shard-rank", type=int, default=0, help="with --shard-of: which rank's shard to render")
    ap.add_argument("--depth", type=int, default=-1, help="override the workload's max ray depth (analysis runs; not the BASELINE config)")
    ap.add_argument("--no-wakeup", action="store_true", help="skip the untimed device wake-up (profiling runs: every k_trace "
                    "launch rocprofv3 sees then belongs to the timed region)")
    ap.add_argument("--no-stats-pass", action="store_true", help="skip the extra counted pass that measures V and T")
    ap.add_argument("--parity-seconds", type=float, default=None, help="host-time budget of the parity leg: the timed region's frame against the "
                    "CPU oracle's render of the same passes (default 25; 0 = skip, which is also --quick's default: the line then carries no parity object)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes (the roofline then quotes the committed per-ray "
                    "figures of profiles/traffic.json, scaled by this run's rays, and says so)")
    ap.add_argument("--quick", action="store_true", help="tuning runs: only the timed region (= --cpu-seconds 0 --no-stats-pass --no-pmc --no-converge)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-keep", default="", help="keep the rocprofv3 output of the counter passes in this directory")
    args = ap.parse_args()
    if args.quick:
        args.cpu_seconds, args.no_stats_pass, args.no_pmc, args.no_converge = 0.0, True, True, True
    if args.parity_seconds is None:
        args.parity_seconds = 0.0 if args.quick else 25.0

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Started plainly with --gpus N: this process becomes the launcher.  It starts the N rank processes itself — one per GPU,
        # through torch.distributed.run on 127.0.0.1, exactly as the driver does — BEFORE anything here touches the GPU (a process
        # that holds the GPU never execs or forks workers), relays rank 0's JSON line (the children inherit stdout) and exits with
        # the launcher's code: a rank that fails makes the whole run fail.
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        # (checked before the GPU is touched: a line that says n_gpus = 1 for a run asked to use 8 must never be printed)
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}: start it as `python bench.py --gpus N` (it launches its own "
                         f"ranks) or under torch.distributed.run with --nproc-per-node equal to --gpus")
    pmc = None
    if world_env == 1 and not args.no_pmc and not args.pmc_child and args.shard_of <= 1:
        # child processes under rocprofv3, before this process initialises the GPU (a process that holds the GPU never execs)
        pmc = pmc_legs(args, keep_dir=args.pmc_keep or None)

    import torch
    import torch.distributed as dist

    # stdout carries exactly ONE JSON line: native libraries (RCCL prints a version banner on fd 1 when a communicator is
    # created) are sent to stderr for the whole run, and the line is written to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = world_env
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # HR_BENCH_ONE_DEVICE: rehearsal of the N > 1 path on a one-GPU box (NOT a benchmark; the line says so): every rank renders its
    # shard on device 0, and because RCCL refuses two ranks on one device the process group is gloo and the tile gather is staged
    # through host memory.  Everything else — launcher, rank environment, sharding, per-batch exchange, assembly, checks — is the
    # code the real run executes.
    one_device = bool(os.environ.get("HR_BENCH_ONE_DEVICE")) and world > 1
    if one_device:
        local_rank = 0
    elif world > 1 and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} device(s) visible (HR_BENCH_ONE_DEVICE=1 rehearses the "
                         f"path on one device; its line is marked as a rehearsal)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # HR_BENCH_FORCE_EXCHANGE: the exchange code path (process group, per-step tile gather, final assembly) in a single process
    forced = bool(os.environ.get("HR_BENCH_FORCE_EXCHANGE")) and world == 1
    exchange = world > 1 or forced
    backend = "gloo" if one_device else "nccl"
    if forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29655")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    elif world > 1:
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        if dist.get_world_size() != args.gpus or dist.get_rank() != rank:
            raise SystemExit(f"bench.py: the communicator has {dist.get_world_size()} rank(s), --gpus says {args.gpus}")
    red_dev = dev if backend == "nccl" else torch.device("cpu")  # where the few scalars reduced over the ranks live

    emulated = args.shard_of > 1 and world == 1
    eng_world = args.shard_of if emulated else world
    passes_total = args.warmup + args.steps
    sc = build_scene(args.workload, args.width, args.height, max(32, passes_total))
    if args.depth >= 0:
        sc.options.max_ray_depth = args.depth
    sc.options.estimator = {"reference": ffi.HR_ESTIMATOR_REFERENCE, "env_mis": ffi.HR_ESTIMATOR_ENV_MIS, "all_lights": ffi.HR_ESTIMATOR_ALL_LIGHTS}[args.estimator]
    stream = torch.cuda.current_stream().cuda_stream
    eng_rank = args.shard_rank if emulated else rank
    # HIP events around every kernel (the contract's live measurement of the dominant kernel's launch duration) at N = 1; a tile
    # shard's launches are short and dependent, and every event record is a packet of 4-8 us between them (2.4 % of a 1/8 shard's
    # 20-pass run, profiles/r4e_shard_ab.txt), so with N > 1 k_trace is timed by the device clock it reads itself (always on;
    # the N = 1 line carries both, they agree).  HR_BENCH_TIME_KERNELS=0/1 forces either.
    time_kernels = os.environ.get("HR_BENCH_TIME_KERNELS", "1" if (world == 1 and not emulated) else "0") != "0"
    TILE = int(os.environ.get("HR_BENCH_TILE", "32"))  # (tile size experiments; SURVEY 8e: 32)
    eng = core.create_engine(device_id=local_rank, rank=eng_rank, world=eng_world, tile_size=TILE, stream=stream, time_kernels=time_kernels)
    sc.apply(eng)  # tables and LUT are generated on the device
    info = eng.scene_info()
    fb = torch.zeros((sc.height, sc.width, 4), dtype=torch.float32, device=dev)
    eng.bind_external_frame(fb.data_ptr())
    gatherer = tiles.FrameGatherer(sc.width, sc.height, rank, world, dev, tile=TILE, dst=0, n_buffers=3, engine=eng, host_staged=one_device) if exchange else None

    # libhrcore injects (and therefore resolves) passes in batches (hr_frame_pass_batch): the accumulation buffer changes once per
    # batch, so that is the exchange cadence as well.
    owned_px = len(tiles.owned_tiles(sc.width, sc.height, eng_rank, eng_world, TILE)) * TILE * TILE
    post_every = max(1, eng.pass_batch(sc.options.max_ray_depth))

    # the per-pass uniform blocks (what PassGenerator::runRenderFrameJob fills in C++, PassGenerator.cpp:349-369) are prepared
    # ahead: in the timed loop the host only hands them over, as the reference's host does
    pass_blocks = [sc.options.pass_params(i) for i in range(passes_total)]

    def step(i):
        eng.render_pass(pass_blocks[i])
        if exchange and (i + 1) % post_every == 0:
            # Progressive display: every step each rank packs the pixels it owns (1/world of the RGBA32F buffer) and
            # RCCL gathers them on rank 0, on a side stream so the exchange overlaps the next pass's kernels.  The
            # buffer holds every pass whose last stage has run (passes still in the pipeline live in their own pass
            # buffers), so any point of the stream gives a consistent progressive image.
            gatherer.post(fb)

    # Untimed wake-up before the W warm-up steps: ONE FILL OF THE PASS PIPELINE — (depth + 2) stages x `post_every` passes per macro step,
    # plus one batch — so that the library's ray arenas have reached their steady-state size (they grow on demand while the pipeline
    # fills: each step carries one more generation of passes; hr_core.hip, Group::arena) and the clocks are up.  132 passes (0.26 s)
    # for c3; rounds 2-3 ran 512 passes / 1 s here to touch 53 GB of per-slot queues.  HR_BENCH_WAKE_MAX overrides.
    t_wake = time.perf_counter()
    n_wake = 0
    wake_cap = int(os.environ.get("HR_BENCH_WAKE_MAX", str(max(64, (sc.options.max_ray_depth + 3) * post_every))))
    while not args.no_wakeup and n_wake < wake_cap:
        eng.render_pass(sc.options.pass_params(n_wake % passes_total))
        n_wake += 1
        if n_wake == 3 * post_every and "HR_BENCH_WAKE_MAX" not in os.environ:
            # by now the library's packet selector has probed this scene and camera; where it sends the camera rays as packets the batch is
            # the neighbouring power of two (12 -> 16 passes per step): the pipeline's fill follows.  (The exchange cadence does not: it is
            # a count of collectives and must be the same on every rank whatever each rank's probe found on its tiles.)
            wake_cap = max(wake_cap, (sc.options.max_ray_depth + 3) * max(1, eng.pass_batch(sc.options.max_ray_depth)))
    eng.flush()
    torch.cuda.synchronize()
    wake_s = time.perf_counter() - t_wake
    eng.clear()
    for i in range(args.warmup):
        step(i)
    eng.clear()  # resets the accumulation buffer, the device counters and the kernel timers
    torch.cuda.synchronize()
    if exchange:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    eng.flush()  # the pass pipeline keeps depth+2 passes in flight: enqueue their remaining stages
    full = fb
    if exchange:
        gatherer.post(fb)  # the finished image
        full = gatherer.finish()
    torch.cuda.synchronize()
    if exchange:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    st = eng.stats()
    kt = eng.kernel_times()
    trace_clock = kt.pop("trace_clock")
    camera_packets = kt.pop("camera_packets")  # (the packet selector's state: camera rays as 64-ray packets or one per lane; its probe's union factor)
    if not time_kernels:  # (no events recorded: the device clock's figure stands in for k_trace's)
        kt["trace"] = trace_clock
    rays = torch.tensor([float(st.rays_closest + st.rays_any), float(st.paths), float(st.rays_closest)], dtype=torch.float64, device=red_dev)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    if exchange:
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    total_rays, total_paths, total_closest = (float(x) for x in rays.tolist())

    if rank == 0:
        # sanity of the timed result itself: every owned pixel got exactly `steps` samples
        a = full[..., 3]
        if emulated:
            a = a[torch.from_numpy(tiles.owner_map(sc.width, sc.height, eng_world, TILE) == eng_rank).to(dev)]
        assert bool((a == float(args.steps)).all()), "sample count mismatch in the accumulation buffer"
        assert bool(torch.isfinite(full).all())
        # digest of the final assembled RGBA32F frame: the same passes give the same bits for every N (SURVEY 8e acceptance:
        # "8-GPU buffer == 1-GPU buffer bit for bit"), so the driver's SCALE lines can be compared with its BENCH line
        import hashlib
        frame_host = full.cpu().numpy()
        frame_sha256 = None if emulated else hashlib.sha256(frame_host.tobytes()).hexdigest()
        # SURVEY 8d metric 1 asks for min / median per pass beside the mean.  The renderer is a pipeline: a macro step advances every
        # in-flight pass by one stage, and a step in which the pipeline is FULL (stages x batch passes in flight) does one pass's worth
        # of work per pass it injects.  libhrcore logs every step's k_trace launch by the device clock (hr_get_step_log): the period
        # between consecutive full steps / passes injected = time per pass in steady state; a run shorter than two pipeline depths
        # has no such steps (the driver's 20 passes: fill and drain only) and reports null with the step log's summary.
        log = eng.step_log()
        batch_stats = None
        if log:
            # (records come sorted by start with their pipeline group: the period is taken between consecutive steps of ONE group, and
            # with G groups stepping in turn a group's period covers the passes all G of them injected meanwhile)
            full_depth = max(r[2] for r in log)
            groups = sorted({r[4] for r in log})
            per_pass = []
            for gr in groups:
                lg = [r for r in log if r[4] == gr]
                per_pass += [(lg[i + 1][0] - lg[i][0]) / (lg[i][3] * len(groups)) for i in range(len(lg) - 1)
                             if lg[i][2] == full_depth and lg[i][3] > 0 and lg[i + 1][2] == full_depth]
            tr = sorted(r[1] for r in log)
            batch_stats = {"macro_steps": len(log), "passes_in_flight_max": full_depth, "pipeline_groups": len(groups), "steady_state_steps": len(per_pass),
                           "k_trace_launch_ms": {"min": tr[0], "median": float(np.median(tr)), "max": tr[-1]},
                           "min": min(per_pass) if len(per_pass) >= 2 else None, "median": float(np.median(per_pass)) if len(per_pass) >= 2 else None,
                           "max": max(per_pass) if len(per_pass) >= 2 else None,
                           "definition": "period between consecutive macro steps with a full pipeline / passes injected per step (device clock, hr_get_step_log); "
                                         "null when the timed region holds fewer than two such steps (it is then fill and drain only)"}
        # ---- parity of the timed region's own output against the CPU oracle (every N; rank 0 holds the assembled frame)
        parity = None
        if args.parity_seconds > 0 and not emulated and not args.pmc_child:
            parity = parity_leg(sc, frame_host, args.warmup, args.steps, args.parity_seconds,
                                f" (assembled from {world} ranks' tiles)" if world > 1 else "")
            assert parity["rel_l2"] <= 1e-4, f"GPU frame differs from the oracle's: {parity}"

        # ---- cpu_baseline leg (N = 1 only): also yields V, T of the roofline model measured by the oracle on the spec BVH
        cpu = None
        if world == 1 and args.cpu_seconds > 0:
            lut, _ = eng.generate_multiscatter_lut()
            cpu = cpu_baseline(sc, args.cpu_seconds, lut)

        # ---- the product's own traversal counters (4-wide quantised BVH), one extra counted pass outside the timed region
        gpu_counts = None
        if not args.no_stats_pass:
            se = core.create_engine(device_id=local_rank, rank=eng_rank, world=eng_world, tile_size=TILE, stream=stream, collect_stats=True)
            sc.apply(se)
            se.render_pass(sc.options.pass_params(args.warmup))
            ss = se.stats()
            gpu_counts = {"node4_visits_per_closest_ray": (ss.node_visits - ss.node_visits_any) / max(ss.rays_closest, 1),
                          "tri_tests_per_closest_ray": (ss.tri_tests - ss.tri_tests_any) / max(ss.rays_closest, 1),
                          "node4_visits_per_occlusion_ray": ss.node_visits_any / max(ss.rays_any, 1),
                          "tri_tests_per_occlusion_ray": ss.tri_tests_any / max(ss.rays_any, 1),
                          "shaded_hit_fraction_H": ss.shaded_hits / max(ss.rays_closest + ss.rays_any, 1),
                          "accumulate_fraction_A": ss.accumulates / max(ss.rays_closest + ss.rays_any, 1)}
            se.close()

        # ---- roofline of the dominant kernel: k_trace (closest-hit + occlusion traversal in one launch).
        # (1) `frac`: COUNTER-BASED.  HBM-side bytes per ray of this configuration (rocprofv3 FETCH_SIZE + WRITE_SIZE of the counter
        #     passes run above on the same command, corrected as MI355X_MICROARCH.md's HBM section prescribes and as calibrated on this
        #     kernel's access pattern, profiles/*_calib_fetch.json: scattered 64-B gathers are counted exactly, the coalesced 48-B/ray
        #     queue stream at half its bytes) x the rays of the timed region / the WALL time of the timed region / 8 TB/s.  The counter
        #     sits on the L2's memory side, so Infinity-Cache hits are included: an upper bound of what reaches HBM, and <= 1 by physics.
        # (2) `valu`: second ceiling, wave-level VALU instructions (SQ_INSTS_VALU) / wall time against 256 CUs x 4 SIMDs x f / 2.
        # (3) `spec_model`: SURVEY §8d's ALGORITHMIC bytes (48 B ray + 16 B result + 64 B x V + 48 B x T with V, T measured by the CPU
        #     oracle on the spec structure: binary LBVH, 64-B two-box nodes, <= 4 triangles per leaf) / the average k_trace launch
        #     duration.  NOT a ceiling for this implementation: the product walks a compressed 4-wide tree whose upper levels stay in
        #     cache, so it moves far fewer bytes than the spec structure would and this ratio can exceed 1.
        ms_trace, n_trace = kt["trace"]
        roofline = None
        wall_s = elapsed
        run_rays = float(st.rays_closest + st.rays_any)
        if n_trace:
            avg_ms = ms_trace / n_trace
            src = None
            per_ray = per_ray_all = valu_per_ray = None
            if pmc and not pmc.get("failed"):
                # the traversal: k_trace, and with it k_raygen_packets when the selector sends the camera rays that way (ray generation fused in)
                kt_own = pmc["kernels"].get("k_trace", {})
                kt_c = dict(kt_own)
                for ck, cv_ in pmc["kernels"].get("k_raygen_packets", {}).items():
                    kt_c[ck] = kt_c.get(ck, 0.0) + cv_
                rays_f, rays_w = pmc["passes"]["fetch"]["rays"], pmc["passes"]["write"]["rays"]
                if "FETCH_SIZE" in kt_c and "WRITE_SIZE" in kt_c:
                    # FETCH_SIZE / WRITE_SIZE are in KiB; + half of the streamed queue reads (48 B per ray) that FETCH_SIZE under-counts
                    per_ray = kt_c["FETCH_SIZE"] * 1024.0 / rays_f + kt_c["WRITE_SIZE"] * 1024.0 / rays_w + 0.5 * 48.0
                    per_ray_all = (sum(k.get("FETCH_SIZE", 0.0) for k in pmc["kernels"].values()) * 1024.0 / rays_f +
                                   sum(k.get("WRITE_SIZE", 0.0) for k in pmc["kernels"].values()) * 1024.0 / rays_w)
                    src = (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU, child passes of this command (wake-up, {args.warmup} warm-up steps, "
                           f"{args.steps} counted steps: only the dispatches of the child's timed region are summed), {pmc['seconds']:.0f} s")
                sq = pmc["passes"].get("sq", {})
                if "rays" in sq and "SQ_INSTS_VALU" in kt_c:
                    valu_per_ray = {"k_trace": kt_c["SQ_INSTS_VALU"] / sq["rays"],
                                    "all_kernels": sum(k.get("SQ_INSTS_VALU", 0.0) for k in pmc["kernels"].values()) / sq["rays"]}
            if per_ray is None:  # counter passes unavailable: the committed measurement of the same workload, scaled by this run's rays
                tpath = os.path.join(ROOT, "profiles", "traffic.json")
                if os.path.exists(tpath):
                    tj = json.load(open(tpath)).get(args.workload)
                    if tj and not tj.get("k_trace_hbm_bytes_per_ray") and tj.get("streamed_queue_read_bytes_per_launch"):
                        tj["k_trace_hbm_bytes_per_ray"] = tj["k_trace_hbm_bytes_per_launch"] / (tj["streamed_queue_read_bytes_per_launch"] / 48.0)
                    if tj and tj.get("k_trace_hbm_bytes_per_ray"):
                        per_ray = tj["k_trace_hbm_bytes_per_ray"]
                        valu_per_ray = tj.get("valu_wave_instructions_per_ray")
                        src = "profiles/traffic.json (committed counter passes of the same workload), scaled by this run's rays — NOT measured in this run"
            if per_ray is not None:
                traffic_run = per_ray * run_rays
                achieved = traffic_run / wall_s / 1e9
                roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                            "traffic": traffic_run / n_trace, "kernel": "k_trace" if not camera_packets[0] else "k_trace + k_raygen_packets (the traversal: camera rays as packets, the rest one ray per lane)",
                            "definition": "counter-based: (FETCH_SIZE + WRITE_SIZE + half the streamed 48 B/ray queue reads) per ray x rays of the timed "
                                          "region / wall time of the timed region; includes Infinity-Cache hits (upper bound of HBM bytes)",
                            "traffic_source": src, "hbm_side_bytes_per_ray": per_ray, "wall_ms": wall_s * 1e3,
                            "avg_launch_ms": avg_ms, "launches": n_trace, "rays_per_launch": (run_rays - (float(st.paths) if camera_packets[0] else 0.0)) / n_trace,
                            "camera_packet_launches": None if not (camera_packets[0] and kt["raygen"][1]) else {
                                "kernel": "k_raygen_packets", "event_bucket_raygen_ms": kt["raygen"][0], "event_bucket_raygen_launches": kt["raygen"][1], "camera_rays": float(st.paths),
                                "note": "ray generation and the camera rays' traversal in one kernel; where it runs beside k_trace on a second stream the event bucket holds only "
                                        "the time it outlasts k_trace (its own duration: the rocprofv3 kernel stats); avg_launch_ms / launches / rays_per_launch above are k_trace's own"},
                            "avg_launch_ms_source": "HIP events on the kernel's stream" if time_kernels else "device clock read inside k_trace (no events on the stream)",
                            "avg_launch_ms_device_clock": trace_clock[0] / max(trace_clock[1], 1),
                            "kernel_time_over_wall_time": sum(v[0] for v in kt.values()) / (wall_s * 1e3)}
                if pmc and not pmc.get("failed") and kt_own.get("FETCH_SIZE") is not None and kt_own.get("WRITE_SIZE") is not None and trace_clock[1]:
                    # the dominant kernel ALONE, as SURVEY 8(d) words it: k_trace's own counter bytes per launch / its own average launch duration
                    own_bytes = (kt_own["FETCH_SIZE"] * 1024.0 / max(kt_own.get("launches_FETCH_SIZE", n_trace), 1) + kt_own["WRITE_SIZE"] * 1024.0 / max(kt_own.get("launches_WRITE_SIZE", n_trace), 1))
                    own_ms = trace_clock[0] / trace_clock[1]
                    roofline["frac_k_trace"] = own_bytes / (own_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                    roofline["k_trace_alone"] = {"bytes_per_launch": own_bytes, "avg_launch_ms": own_ms, "gbs": own_bytes / (own_ms * 1e-3) / 1e9,
                                                 "definition": "k_trace's own FETCH_SIZE + WRITE_SIZE (KiB -> bytes) per launch of the counter passes' timed region / the average k_trace launch "
                                                               "of THIS run's timed region by the kernel's own device clock / 8 TB/s; Infinity-Cache hits included (an upper bound of HBM bytes)"}
                if per_ray_all:
                    roofline["all_kernels_frac"] = per_ray_all * run_rays / wall_s / 1e9 / HBM_PEAK_GBS
                if valu_per_ray:
                    cal = os.path.join(ROOT, "profiles", "r2_calib_valu.json")
                    per_simd = VALU_CLOCK_GHZ / VALU_CYCLES_PER_WAVE_OP  # G wave-instructions / s / SIMD
                    peak_src = f"{VALU_CLOCK_GHZ} GHz / {VALU_CYCLES_PER_WAVE_OP:g} cycles per wave64 op (MI355X_MICROARCH.md)"
                    if os.path.exists(cal):
                        runs = json.load(open(cal)).get("runs", [])
                        if runs:
                            per_simd = max(r["per_simd_ginstr_per_s"] for r in runs)
                            peak_src = "tools/calib_valu.hip, measured (profiles/r2_calib_valu.json)"
                    n_cus = int(getattr(torch.cuda.get_device_properties(dev), "multi_processor_count", 256))
                    peak = n_cus * 4 * per_simd  # G wave-instructions / s
                    ach = valu_per_ray["all_kernels"] * run_rays / wall_s / 1e9
                    roofline["valu"] = {"bound": "valu_issue", "achieved": ach, "peak": peak, "unit": "G wave-instructions/s", "frac": ach / peak,
                                        "peak_source": peak_src, "wave_instructions_per_ray": valu_per_ray,
                                        "note": "issue-rate ceiling of plain FMAs at the calibrated clock; the instruction mix of k_trace occupies a SIMD "
                                                "for 3.3-4 cycles per instruction (tools/calib_ops.hip) at a shader clock of ~2.3 GHz (tools/tailprof.py): see `units` for busy fractions"}
                if pmc and not pmc.get("failed") and "rays" in pmc["passes"].get("units", {}):
                    ku = dict(pmc["kernels"].get("k_trace", {}))  # (the dominant kernel; the packet kernel, when in use, is listed beside it)
                    if ku.get("GRBM_GUI_ACTIVE") and ku.get("TA_TA_BUSY_sum") is not None:
                        n_xcd = 8.0   # GRBM_GUI_ACTIVE is reported summed over the XCDs; TA counters over the CUs; SQ_ACTIVE_* in quad-cycles
                        n_cus = float(getattr(torch.cuda.get_device_properties(dev), "multi_processor_count", 256))
                        cyc = ku["GRBM_GUI_ACTIVE"] / n_xcd
                        # VALU busy: SQ_ACTIVE_INST_VALU charges every VALU wave-instruction one quad-cycle whatever it is (it equals
                        # SQ_INSTS_VALU to 0.3 %), and x 4 read as "busy cycles" exceeded 1 for the FMA-heavy packet kernel.  The
                        # instructions are priced by issue class instead — the SQ's typed instruction counters of the same counter pass give
                        # the classes — with the LOWEST cost tools/calib_ops.hip measured for a class on a saturated SIMD (ns per wave64
                        # instruction: f32 add / mul / fma 0.94, transcendentals 3.41, everything else 1.71; profiles/r4u_calib_ops.json), over
                        # the SIMD-time the kernel had in THAT pass (its dispatches' durations by the profiler's timestamps x 4 SIMDs x CUs):
                        # a lower bound of the issue time the instructions need, so the fraction cannot exceed 1.
                        # (the typed counters know f32 add / mul / fma and transcendentals; of the REST — conversions, min / max, compares, integer
                        # and bit work — a kernel-specific share still issues at the double rate (v_mov, v_and, v_add_u32, v_cndmask_e32): taken from
                        # the kernel's own ISA, tools/valu_mix.py -> profiles/r5_valu_mix.json; 0 when that file is missing: the rest at the single rate)
                        try:
                            mix = json.load(open(os.path.join(ROOT, "profiles", "r5_valu_mix.json")))["kernels"]
                        except (OSError, ValueError, KeyError):
                            mix = {}

                        def valu_busy_of(kd, kn=""):
                            n = kd.get("SQ_INSTS_VALU", 0.0)
                            fast = kd.get("SQ_INSTS_VALU_ADD_F32", 0.0) + kd.get("SQ_INSTS_VALU_MUL_F32", 0.0) + kd.get("SQ_INSTS_VALU_FMA_F32", 0.0)
                            trans = kd.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
                            dur = kd.get("duration_ns_sq", 0.0)
                            if not n or not dur:
                                return None
                            share = float(mix.get(kn, {}).get("double_rate_share_of_rest", 0.0))
                            issue_ns = 0.9375 * fast + 3.4082 * trans + (0.9375 * share + 1.7121 * (1.0 - share)) * max(n - fast - trans, 0.0)
                            return {"busy": issue_ns / (n_cus * 4.0) / dur, "wave_instructions": n, "double_rate_share": fast / n, "transcendental_share": trans / n,
                                    "double_rate_share_of_rest_static": share,
                                    "ns_per_instruction": issue_ns / n, "kernel_ms_in_the_counter_pass": dur * 1e-6,
                                    "lane_utilisation": (kd["SQ_THREAD_CYCLES_VALU"] / (64.0 * n)) if kd.get("SQ_THREAD_CYCLES_VALU") else None}
                        vb = {kn: valu_busy_of(kd, kn) for kn, kd in pmc["kernels"].items() if kn in ("k_trace", "k_raygen_packets", "k_shade_hit", "k_shade_sort")}
                        roofline["units"] = {
                            "kernel": "k_trace", "kernel_cycles": cyc,
                            "note": None if not camera_packets[0] else "camera rays travel as packets in k_raygen_packets (per_kernel): VALU-bound, addressers idle — the complement of "
                                    "k_trace, beside which it runs on a second stream in the timed run; the counter passes serialise the two, so each line describes its kernel alone",
                            "per_kernel": {kn: {"cycles": kd["GRBM_GUI_ACTIVE"] / n_xcd, "ta_busy": kd.get("TA_TA_BUSY_sum", 0.0) / n_cus / (kd["GRBM_GUI_ACTIVE"] / n_xcd),
                                                "valu_busy": (vb.get(kn) or {}).get("busy"), "valu": vb.get(kn)}
                                           for kn, kd in pmc["kernels"].items() if kn in ("k_trace", "k_raygen_packets", "k_shade_hit", "k_shade_sort") and kd.get("GRBM_GUI_ACTIVE")},
                            "ta_busy": ku["TA_TA_BUSY_sum"] / n_cus / cyc,
                            "valu_busy": (vb.get("k_trace") or {}).get("busy"),
                            "ta_wave_loads_per_ray": ku.get("TA_FLAT_READ_WAVEFRONTS_sum", 0.0) / pmc["passes"]["units"]["rays"],
                            "definition": "ta_busy: busy cycles of the CU's texture addresser (TA_TA_BUSY_sum / CUs) over the cycles the kernel ran; "
                                          "valu_busy: VALU wave-instructions priced by issue class (0.94 ns x f32 add/mul/fma + 3.41 x transcendentals + 1.71 x the rest, of which the "
                                          "kernel's static share of double-rate integer / move instructions at 0.94: costs from tools/calib_ops.hip, classes from the SQ's typed "
                                          "counters and tools/valu_mix.py) over the kernel's SIMD-time in the same counter pass; cycles: "
                                          "over the cycles k_trace ran (GRBM_GUI_ACTIVE / 8 XCDs), summed over the k_trace launches of the TIMED region of a counter pass "
                                          "that runs this command's own configuration (wake-up, warm-up, same step count: a warm device): "
                                          "the two units that bind the kernel (scattered 16-B-per-lane loads cost one TA cycle per lane and instruction, "
                                          "tools/calib_tcp.hip)"}
            vt = cpu
            if vt is None:
                vpath = os.path.join(ROOT, "profiles", "vt_spec.json")
                if os.path.exists(vpath):
                    vt = json.load(open(vpath)).get(args.workload)
            if vt is not None and roofline is not None:
                bytes_closest = 48.0 + 16.0 + 64.0 * vt["V"] + 48.0 * vt["T"]
                bytes_any = 48.0 + 16.0 + 64.0 * vt["V_any"] + 48.0 * vt["T_any"]
                total_bytes = bytes_closest * float(st.rays_closest) + bytes_any * float(st.rays_any)
                roofline["spec_model"] = {
                    "note": "SURVEY 8d algorithmic bytes on the SPEC structure (binary LBVH); not a ceiling for a compressed 4-wide tree — may exceed 1",
                    "algorithmic_bytes_per_launch": total_bytes / n_trace, "achieved_gbs_per_launch_duration": total_bytes / n_trace / (avg_ms * 1e-3) / 1e9,
                    "ratio_to_hbm_peak": total_bytes / n_trace / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "algorithmic_bytes_per_closest_ray": bytes_closest, "algorithmic_bytes_per_occlusion_ray": bytes_any,
                    "V": vt["V"], "T": vt["T"], "V_any": vt["V_any"], "T_any": vt["T_any"],
                    "vt_source": "cpu oracle, this run" if vt is cpu else "profiles/vt_spec.json"}
            if gpu_counts is not None and roofline is not None:
                # the same accounting on the tree the product really walks: 64-B 4-wide nodes, 48-B triangles (every visit charged)
                b = (48.0 + 16.0) * run_rays + 64.0 * (gpu_counts["node4_visits_per_closest_ray"] * st.rays_closest + gpu_counts["node4_visits_per_occlusion_ray"] * st.rays_any) \
                    + 48.0 * (gpu_counts["tri_tests_per_closest_ray"] * st.rays_closest + gpu_counts["tri_tests_per_occlusion_ray"] * st.rays_any)
                roofline["product_tree_model"] = {"note": "algorithmic bytes of the tree the kernel walks (every node visit = 64 B, every triangle test = 48 B, "
                                                          "no cache credit) / wall time", "bytes_per_ray": b / run_rays, "gbs": b / wall_s / 1e9,
                                                  "ratio_to_hbm_peak": b / wall_s / 1e9 / HBM_PEAK_GBS}

        # the display resolve, outside the timed region: device-side displayGL.frag -> RGBA8
        disp_ms = None
        if not emulated:
            shown = torch.empty((sc.height, sc.width), dtype=torch.int32, device=dev)
            P = ffi.display_params(tonemapping_enabled=True)
            eng.display_device(shown.data_ptr(), P, ffi.HR_DISPLAY_RGBA8)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                eng.display_device(shown.data_ptr(), P, ffi.HR_DISPLAY_RGBA8)
            e1.record()
            torch.cuda.synchronize()
            disp_ms = e0.elapsed_time(e1) / 10.0
            if world == 1:
                assert int(shown.view(torch.uint8).reshape(sc.height, sc.width, 4)[..., 3].min().item()) == 255

        # BASELINE metric 2, measured live, as defined: p50 over ALL 16 runs (Sobol sequence index 0..15 of the SequenceOffsets table),
        # each against the one 8192-pass reference image of its estimator (~2 min for the reference's estimator on c3, ~1.2 min for
        # HR_ESTIMATOR_ALL_LIGHTS).  --converge-budget bounds the leg's wall time: past it (and after at least 3 runs) the leg stops
        # and the line says over how many runs the statistic is.
        conv = None
        if world == 1 and not emulated and not args.no_converge and not args.pmc_child:
            n_runs = args.converge_runs
            conv = convergence_leg(core, sc, dev, stream, args.converge_cap, n_runs, budget_s=None if args.converge else args.converge_budget)
            conv["measured_live"] = True
            conv["runs_of_16"] = conv["runs_made"]
            conv["estimator"] = args.estimator
            if args.estimator == "reference":
                keys = ("p50", "runs", "runs_made", "median_err_at_pass", "reference_render_s", "runs_s")
                if args.converge_mis:
                    # the same leg with the importance-sampled environment + MIS estimator (its own 8192-pass reference image)
                    sc.options.estimator = ffi.HR_ESTIMATOR_ENV_MIS
                    conv_mis = convergence_leg(core, sc, dev, stream, args.converge_cap, n_runs, budget_s=None if args.converge else args.converge_budget)
                    conv["env_mis"] = {k: conv_mis[k] for k in keys}
                # ... and, in every run, with the opt-in estimator that samples an analytic light AND the environment at every vertex
                # (HR_ESTIMATOR_ALL_LIGHTS: the product's answer to metric 2; the reference-faithful estimator stays the default)
                sc.options.estimator = ffi.HR_ESTIMATOR_ALL_LIGHTS
                conv_all = convergence_leg(core, sc, dev, stream, args.converge_cap, n_runs, budget_s=None if args.converge else 0.6 * args.converge_budget)
                sc.options.estimator = ffi.HR_ESTIMATOR_REFERENCE
                conv["all_lights"] = {k: conv_all[k] for k in keys}
        else:
            cpath = os.path.join(ROOT, "profiles", "converge.json")
            if os.path.exists(cpath):
                conv = json.load(open(cpath)).get(args.workload)
                if conv is not None:
                    conv["measured_live"] = False

        mrays = total_rays / elapsed / 1e6
        out = {
            "metric": "Mrays/s at 1920x1080, 8 bounces" if args.workload in ("c2", "c3", "c3d", "terrain") else f"Mrays/s ({args.workload})",
            "value": mrays, "unit": "Mrays/s", "n_gpus": (dist.get_world_size() if world > 1 else 1), "steps": args.steps, "warmup": args.warmup,
            # untimed device wake-up BEFORE the W warm-up steps (clock ramp / first touch of the pass slots on a freshly started box)
            "wakeup_passes": n_wake, "wakeup_s": wake_s,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_min": batch_stats["min"] if batch_stats else None, "ms_per_step_median": batch_stats["median"] if batch_stats else None,
            "ms_per_step_steady_state": batch_stats,
            "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {WORKLOADS[args.workload]['desc']}", "width": sc.width, "height": sc.height,
                       "max_ray_depth": sc.options.max_ray_depth, "triangles": int(info.n_triangles), "bvh_nodes": int(info.n_nodes),
                       "sharding": f"32x32 pixel tiles round-robin over {world} GPU(s)" + (f"; RCCL gather of the owned RGBA32F tiles to rank 0 every {post_every} step(s) (= every resolved batch of passes), overlapped on a side stream" if exchange else ""),
                       "seed": hex(scenes.SEED)},
            "roofline": roofline,
            "pmc_passes": None if pmc is None else {"seconds": pmc.get("seconds"), "failed": bool(pmc.get("failed")), "passes": pmc.get("passes"),
                                                    "k_trace": pmc.get("kernels", {}).get("k_trace")},
            "cpu_baseline": cpu,
            "parity": parity,
            "frame_sha256": frame_sha256,
            "passes_to_converge": conv,
            "extra": {"rays": total_rays, "paths_per_s": total_paths / elapsed, "closest_rays": total_closest,
                      "rays_per_path": total_rays / max(total_paths, 1.0), "bvh_build_ms": info.build_ms,
                      "camera_rays": {"traced_as": f"packets of {int(camera_packets[2])} passes x {64 // max(int(camera_packets[2]), 1)} pixels" if camera_packets[0] else "one ray per lane",
                                      "packet_union": round(camera_packets[1], 3)},
                      "kernel_ms_rank0": {k: v[0] for k, v in kt.items()}, "kernel_launches_rank0": {k: v[1] for k, v in kt.items()},
                      "display_resolve_ms_rgba8": disp_ms,
                      "gpu_traversal_counters": gpu_counts},
        }
        if one_device:
            out["rehearsal_one_device"] = True
            out["metric"] += f" [REHEARSAL: {world} ranks on ONE device over gloo, not a benchmark result]"
        if emulated:
            out["emulated_shard_of"] = eng_world
            out["metric"] += f" [EMULATED rank {eng_rank} of {eng_world}, not a benchmark result]"
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if exchange:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
