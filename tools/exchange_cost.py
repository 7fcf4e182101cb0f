import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist
from heatray_amd import tiles
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29688")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
saved = os.dup(1); os.dup2(2, 1)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
fb = torch.rand((1080, 1920, 4), device=dev)
for nb, ov in ((3, True), (3, False), (8, True)):
    g = tiles.FrameGatherer(1920, 1080, 0, 1, dev, n_buffers=nb, overlap=ov)
    for _ in range(5): g.post(fb)
    g.finish(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): g.post(fb)
    g.finish(); torch.cuda.synchronize()
    print("n_buffers", nb, "overlap", ov, "ms per post", (time.perf_counter() - t0) / 50 * 1e3, file=sys.stderr)
# pieces
idx = g.idx[0]; send = g.send[0]; flat = g.full.view(-1, 4)
def timeit(f, n=50):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("index_select", timeit(lambda: torch.index_select(fb.view(-1, 4), 0, idx, out=send)), file=sys.stderr)
print("gather", timeit(lambda: dist.gather(send, g.recv[0], dst=0)), file=sys.stderr)
print("index_copy", timeit(lambda: flat.index_copy_(0, idx, g.recv[0][0])), file=sys.stderr)
dist.destroy_process_group()
