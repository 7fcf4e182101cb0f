"""tools/mem_probe.py [workload] [passes] [budget GiB] — GPU box: device memory the library holds after a run of `passes` passes (hipMemGetInfo
through torch), its throughput, and — with a budget (hr_ctx_desc.memory_budget) — the batch the library chose under it, before and after it had
seen a full pipeline."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from heatray_amd import core
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 256
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
torch.cuda.init()
free0, total = torch.cuda.mem_get_info()
sc = bench.build_scene(wl, 0, 0, max(32, passes))
e = core.create_engine(stream=torch.cuda.current_stream().cuda_stream, memory_budget=int(budget * 2**30))
sc.apply(e)
free1, _ = torch.cuda.mem_get_info()
b0 = e.pass_batch(sc.options.max_ray_depth)
peak = 0
t0 = time.perf_counter()
for i in range(passes):
    e.render_pass(sc.options.pass_params(i))
    if i % 16 == 15:
        peak = max(peak, free1 - torch.cuda.mem_get_info()[0])
e.flush(); e.synchronize()
el = time.perf_counter() - t0
free2, _ = torch.cuda.mem_get_info()
peak = max(peak, free1 - free2)
st = e.stats()
print(f"{wl}: {sc.width}x{sc.height}, {passes} passes" + (f", budget {budget:.1f} GiB" if budget else ", no budget") +
      f": scene + tables {(free0 - free1) / 2**30:.2f} GiB, pass pipeline (pass buffers + ray arenas) {(free1 - free2) / 2**30:.2f} GiB (peak {peak / 2**30:.2f}), "
      f"total {(free0 - free2) / 2**30:.2f} GiB of {total / 2**30:.0f}; passes per step {b0} -> {e.pass_batch(sc.options.max_ray_depth)}; "
      f"{(st.rays_closest + st.rays_any) / el / 1e6:.0f} Mrays/s incl. the pipeline's first fill; rays/path {(st.rays_closest + st.rays_any) / max(st.paths, 1):.2f}")
