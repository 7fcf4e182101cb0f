"""tools/r4_shard_steps.py — GPU box: the macro steps of one rank's 1/8 shard of c3 over 20 passes (step log, kernel buckets): where a shard's time goes."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from heatray_amd import core
sc = bench.build_scene("c3", 0, 0, 64)
e = core.create_engine(rank=0, world=8, tile_size=32, time_kernels=True)
sc.apply(e)
for rep in range(3):
    e.clear()
    for i in range(132): e.render_pass(sc.options.pass_params(i % 64))
    e.flush(); e.synchronize()
e.clear()
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(20): e.render_pass(sc.options.pass_params(5 + i))
e.flush(); e.synchronize()
el = time.perf_counter() - t0
print("20 passes: %.3f ms (%.4f ms/step)" % (el * 1e3, el * 1e3 / 20))
print("kernel ms", {k: (round(v[0], 3), v[1]) if isinstance(v, tuple) and len(v) == 2 else v for k, v in e.kernel_times().items()})
for r in e.step_log(): print("  step start %.3f ms  k_trace %.3f ms  in flight %d injected %d" % r[:4])
