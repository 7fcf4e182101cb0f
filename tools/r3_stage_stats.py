"""Rays and node visits per pipeline stage of a workload (differences of runs with max depth 0, 1, 2, ...).  GPU box."""
import os, sys
sys.path.insert(0, os.getcwd())
import bench
from heatray_amd import core
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
sc = bench.build_scene(wl, 0, 0, 64)
top = sc.options.max_ray_depth
prev = None
for depth in range(0, top + 1):
    sc.options.max_ray_depth = depth
    eng = core.create_engine(collect_stats=True)
    sc.apply(eng)
    for i in range(4): eng.render_pass(sc.options.pass_params(i))
    s = eng.stats()
    cur = dict(c=s.rays_closest / 4, a=s.rays_any / 4, vc=(s.node_visits - s.node_visits_any) / 4, va=s.node_visits_any / 4)
    p = prev or dict(c=0, a=0, vc=0, va=0)
    dc, da = cur["c"] - p["c"], cur["a"] - p["a"]
    print(f"stage {depth}: closest {dc/1e3:9.1f}k (visits/ray {(cur['vc']-p['vc'])/max(dc,1):6.1f})   occlusion emitted by its hits {da/1e3:9.1f}k (visits/ray {(cur['va']-p['va'])/max(da,1):6.1f})")
    prev = cur
    eng.close()
