"""Fullness of the 4-wide tree of a bench workload: children per node, inner / triangle children, by level (reads the tree cache file the
library writes).   python tools/tree_stats.py [workload]      (GPU box)"""
import os, struct, sys, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from heatray_amd import core

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
sc = bench.build_scene(wl, 0, 0, 32)
path = os.path.join(tempfile.mkdtemp(dir="/tmp"), "t.hrbvh")
eng = core.create_engine()
eng.set_scene_cache(path)
sc.apply(eng)
eng.render_pass(sc.options.pass_params(0))
eng.readback()
raw = open(path, "rb").read()
magic, version, node_bytes, key, n_tris, n_nodes, levels, root_leaf, tri_slots, pad = struct.unpack_from("<8sIIQIIIIII", raw, 0)
KMAX = int(os.environ.get("KMAXLEVELS", "64"))
hdr = 8 + 4 + 4 + 8 + 6 * 4 + (KMAX + 1) * 4
hdr = (hdr + 7) // 8 * 8 + 8
level_start = struct.unpack_from("<%dI" % (KMAX + 1), raw, 8 + 4 + 4 + 8 + 6 * 4)
nodes = np.frombuffer(raw, dtype=np.uint32, count=n_nodes * 16, offset=hdr).reshape(n_nodes, 16)
meta = nodes[:, 3]
n_inner, n_valid = (meta >> 24) & 7, meta >> 27
print(f"{wl}: {n_tris} triangles, {n_nodes} nodes ({n_nodes / max(n_tris - 1, 1) * 3:.2f} x the minimum (n - 1) / 3), {levels} levels, header {hdr} B")
print("children per node:", {int(k): int((n_valid == k).sum()) for k in range(1, 5)}, " mean", float(n_valid.mean()))
leaf_only = n_inner == 0
print("nodes with triangles only:", int(leaf_only.sum()), " their children:", {int(k): int((n_valid[leaf_only] == k).sum()) for k in range(1, 5)})
print("inner children per node:", {int(k): int((n_inner == k).sum()) for k in range(0, 5)})
# share of the expected node visits (surface area = visit probability of a random ray) by level and by kind of node
box_off = hdr + n_nodes * 64
boxes = np.frombuffer(raw, dtype=np.float32, count=n_nodes * 6, offset=box_off).reshape(n_nodes, 6)
ext = np.maximum(boxes[:, 3:6] - boxes[:, 0:3], 0)
area = ext[:, 0] * ext[:, 1] + ext[:, 1] * ext[:, 2] + ext[:, 2] * ext[:, 0]
tot = float(area.sum())
print("share of summed node area: triangles-only nodes %.3f (two triangles %.3f, three %.3f, four %.3f)" % (
    area[leaf_only].sum() / tot, area[leaf_only & (n_valid == 2)].sum() / tot, area[leaf_only & (n_valid == 3)].sum() / tot, area[leaf_only & (n_valid == 4)].sum() / tot))
for L in range(levels):
    a, b = level_start[L], level_start[L + 1]
    if b > a:
        print(f"level {L:2d}: {b - a:7d} nodes, area share {area[a:b].sum() / tot:.3f}")
