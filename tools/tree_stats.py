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
