"""tools/tree_costs.py [workload ...] — GPU box: the two candidate trees' collapse costs (summed 4-wide node area / root area) and which one was kept."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from heatray_amd import core
for wl in sys.argv[1:] or ["c1", "c2", "c2p", "c3", "c3d", "c5", "terrain"]:
    sc = bench.build_scene(wl, 64, 64, 32)
    e = core.create_engine()
    sc.apply(e)
    i = e.scene_info()
    print(f"{wl}: triangles {i.n_triangles}  cost radix {i.cost_radix:.3f}  ploc {i.cost_ploc:.3f}  ratio {i.cost_ploc / max(i.cost_radix, 1e-30):.3f}  kept {'PLOC' if i.builder else 'radix'}  levels {i.bvh_levels}  build {i.build_ms:.2f} ms")
    e.close()
