import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from heatray_amd import core
sc = bench.build_scene("c3", 0, 0, 32)
eng = core.create_engine()
sc.apply(eng)
W, H = 1920, 1080
xs = (np.arange(W, dtype=np.float32) + 0.5) / W * 2.4 - 1.2
ys = (np.arange(H, dtype=np.float32) + 0.5) / H * 1.35 - 0.675
# 32x32 tiles, 8x8 blocks inside: the order k_raygen emits
order = []
gx, gy = np.meshgrid(np.arange(W), np.arange(H))
tile = (gy // 32) * ((W + 31) // 32) + gx // 32
blk = ((gy % 32) // 8) * 4 + (gx % 32) // 8
inb = (gy % 8) * 8 + gx % 8
key = (tile.astype(np.int64) * 16 + blk) * 64 + inb
idx = np.argsort(key.ravel(), kind="stable")
px, py = gx.ravel()[idx], gy.ravel()[idx]
org = np.tile(np.array([0.0, 0.0, 6.0], np.float32), (W * H, 1))
tgt = np.stack([xs[px], ys[py], np.zeros(W * H, np.float32)], axis=1)
d = tgt - org
d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
perm = np.random.default_rng(0).permutation(W * H)
def t(o, dd, name):
    eng.debug_trace(o[:1000], dd[:1000])
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); h = eng.debug_trace(o, dd); best = min(best, time.perf_counter() - t0)
    print(f"{name}: {best*1e3:.2f} ms total (incl. copies) hits {(h['prim']>=0).mean():.3f}")
t(org, d, "tile order (coherent)")
t(org[perm], d[perm], "random order")
# rays from inside the scene, random directions (like bounces)
rng = np.random.default_rng(1)
o2 = rng.uniform(-1, 1, (W * H, 3)).astype(np.float32)
d2 = rng.normal(size=(W * H, 3)); d2 = (d2 / np.linalg.norm(d2, axis=1, keepdims=True)).astype(np.float32)
t(o2, d2, "random interior rays")
