"""tools/r5_soak_vs_oracle.py [workload] [passes] [k] [HR_TUNE] — GPU box: a LONG render against the checker.

The GPU renders the whole frame for `passes` passes; the parity-build oracle (oracle/liboracle.so, -O2 -ffp-contract=off, OpenMP
on the host cores) renders the same passes on the interleaved tile shard k // 2 of k (32x32 tiles over the whole frame), and the two
HDR buffers are compared bit for bit on the pixels that shard owns.  c3, 640 passes, k = 8: 1.7 x 10^8 camera rays plus their
bounces on each side — the pass counts the viewer runs at, where a once-in-6 x 10^8-rays event (a float32
Möller–Trumbore phantom hit on a sliver triangle that one tree's traversal tests and the other's box test turns away) shows up.
Prints one JSON line; differing pixels are listed with both values."""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 640
k = int(sys.argv[3]) if len(sys.argv) > 3 else 8
if len(sys.argv) > 4:
    os.environ["HR_TUNE"] = sys.argv[4]
import numpy as np
import bench, oracle_lib
from heatray_amd import core, tiles

sc = bench.build_scene(wl, int(os.environ.get("W", "0")), int(os.environ.get("H", "0")), passes)
t0 = time.perf_counter()
e = core.create_engine()
sc.apply(e)
for i in range(passes):
    e.render_pass(sc.options.pass_params(i))
gpu = e.readback().copy()
t_gpu = time.perf_counter() - t0
print(f"gpu: {passes} passes in {t_gpu:.1f} s", flush=True)

t0 = time.perf_counter()
o = oracle_lib.engine(rank=k // 2, world=k, tile_size=32) if k > 1 else oracle_lib.engine()
sc.apply(o)
for i in range(passes):
    o.render_pass(sc.options.pass_params(i))
    if i % 32 == 31:
        print(f"oracle: pass {i + 1} of {passes}, {time.perf_counter() - t0:.0f} s", flush=True)  # (a silent command is taken to be hung)
ref = o.readback().copy()
t_ora = time.perf_counter() - t0
own = np.ones(ref.shape[:2], dtype=bool) if k == 1 else (tiles.owner_map(sc.width, sc.height, k) == k // 2)
diff = (gpu != ref).any(axis=-1) & own
ys, xs = np.nonzero(diff)
g, r = gpu[own].astype(np.float64), ref[own].astype(np.float64)
out = {"workload": wl, "passes": passes, "shard": f"{k // 2} of {k}", "pixels": int(own.sum()), "camera_rays": int(own.sum()) * passes,
       "differing_pixels": int(diff.sum()), "rel_l2": float(np.linalg.norm(g - r) / max(np.linalg.norm(r), 1e-30)),
       "gpu_s": t_gpu, "oracle_s": t_ora, "hr_tune": os.environ.get("HR_TUNE", ""),
       "gpu_sha256": hashlib.sha256(np.ascontiguousarray(gpu).tobytes()).hexdigest(),
       "differing": [{"x": int(x), "y": int(y), "gpu": gpu[y, x].tolist(), "oracle": ref[y, x].tolist()} for y, x in list(zip(ys, xs))[:16]]}
print(json.dumps(out))
