"""tools/r4_soak_diff.py [workload] [passes] — GPU box: where do the frames of two trace modes differ after a long run without read-backs?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
from heatray_amd import core
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 600
frames = {}
for tune in (os.environ.get("TUNE_A", "packets=1,corun=0"), os.environ.get("TUNE_B", "packets=0")):
    os.environ["HR_TUNE"] = tune
    sc = bench.build_scene(wl, 0, 0, passes)
    e = core.create_engine()
    sc.apply(e)
    for i in range(passes):
        e.render_pass(sc.options.pass_params(i))
    frames[tune] = e.readback().copy()
    print(tune, "alpha min/max", frames[tune][..., 3].min(), frames[tune][..., 3].max(), flush=True)
    e.close()
a, b = list(frames.values())
d = np.any(a != b, axis=2)
print("differing pixels", int(d.sum()), "of", d.size)
if d.any():
    ys, xs = np.nonzero(d)
    print("rows", ys.min(), ys.max(), "cols", xs.min(), xs.max())
    rel = np.abs(a - b)[d].max() / max(np.abs(a).max(), 1e-30)
    print("max abs diff", np.abs(a - b)[d].max(), "relative to frame max", rel)
    for k in range(min(5, len(ys))):
        print((int(xs[k]), int(ys[k])), a[ys[k], xs[k]], b[ys[k], xs[k]])
