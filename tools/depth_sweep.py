import os, sys, time
sys.path.insert(0, os.getcwd())
import bench
from heatray_amd import core
sc = bench.build_scene("c3", 0, 0, 64)
for depth in (0, 1, 8):
    sc.options.max_ray_depth = depth
    eng = core.create_engine(time_kernels=True)
    sc.apply(eng)
    for i in range(8): eng.render_pass(sc.options.pass_params(i))
    eng.clear()
    t0 = time.perf_counter()
    for i in range(32): eng.render_pass(sc.options.pass_params(i))
    eng.flush(); eng.synchronize()
    el = time.perf_counter() - t0
    st = eng.stats(); kt = eng.kernel_times()
    rays = st.rays_closest + st.rays_any
    print(f"depth {depth}: {rays/el/1e6:8.1f} Mrays/s  {el/32*1e3:.3f} ms/pass  rays/pass {rays/32/1e6:.2f}M  trace {kt['trace'][0]/max(kt['trace'][1],1):.3f} ms x{kt['trace'][1]}  shade {kt['shade'][0]/max(kt['shade'][1],1):.3f}")
    eng.close()
