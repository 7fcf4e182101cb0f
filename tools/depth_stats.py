import os, sys
sys.path.insert(0, os.getcwd())
import bench
from heatray_amd import core
sc = bench.build_scene("c3", 0, 0, 64)
for depth in (0, 1, 2, 8):
    sc.options.max_ray_depth = depth
    eng = core.create_engine(collect_stats=True)
    sc.apply(eng)
    for i in range(4): eng.render_pass(sc.options.pass_params(i))
    s = eng.stats()
    print(f"depth {depth}: closest rays/pass {s.rays_closest/4/1e6:.2f}M  any {s.rays_any/4/1e6:.2f}M  node visits per closest {(s.node_visits-s.node_visits_any)/max(s.rays_closest,1):.1f} per any {s.node_visits_any/max(s.rays_any,1):.1f}  tri tests per closest {(s.tri_tests-s.tri_tests_any)/max(s.rays_closest,1):.2f} per any {s.tri_tests_any/max(s.rays_any,1):.2f}")
    eng.close()
