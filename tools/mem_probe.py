"""tools/mem_probe.py [workload] [passes] — GPU box: device memory the library holds after a run of `passes` passes (hipMemGetInfo through torch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from heatray_amd import core
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 256
torch.cuda.init()
free0, total = torch.cuda.mem_get_info()
sc = bench.build_scene(wl, 0, 0, max(32, passes))
e = core.create_engine(stream=torch.cuda.current_stream().cuda_stream)
sc.apply(e)
free1, _ = torch.cuda.mem_get_info()
for i in range(passes):
    e.render_pass(sc.options.pass_params(i))
e.flush(); e.synchronize()
free2, _ = torch.cuda.mem_get_info()
st = e.stats()
print(f"{wl}: {sc.width}x{sc.height}, {passes} passes: scene + tables {(free0 - free1) / 2**30:.2f} GiB, pass pipeline (pass buffers + ray arenas) {(free1 - free2) / 2**30:.2f} GiB, "
      f"total {(free0 - free2) / 2**30:.2f} GiB of {total / 2**30:.0f}; rays/path {(st.rays_closest + st.rays_any) / max(st.paths, 1):.2f}")
