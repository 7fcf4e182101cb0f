"""The display exchange of the N > 1 bench path on a real GPU: RCCL refuses two ranks on one device, so this is the
world-size-1 rehearsal of exactly the code bench.py runs per step (pack on the compute stream, RCCL gather on the side
stream from rotating staging buffers, assembled frame on rank 0) while the engine keeps rendering on the compute stream."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_frame_gatherer_on_rccl_side_stream(golden):
    import torch
    import torch.distributed as dist
    from heatray_amd import core, scenes, tiles

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 200))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        sc = scenes.multi_material(200, 120, bounces=4)   # 7 x 4 tiles, ragged right and top edges
        stream = torch.cuda.current_stream().cuda_stream
        eng = core.create_engine(device_id=0, rank=0, world=1, tile_size=32, stream=stream)
        sc.apply(eng, lut=golden["multiscatter_lut"])
        fb = torch.zeros((sc.height, sc.width, 4), dtype=torch.float32, device=dev)
        eng.bind_external_frame(fb.data_ptr())
        g = tiles.FrameGatherer(sc.width, sc.height, 0, 1, dev, tile=32, dst=0, n_buffers=3, engine=eng)   # core pack / unpack kernels
        gt = tiles.FrameGatherer(sc.width, sc.height, 0, 1, dev, tile=32, dst=0, n_buffers=2)               # torch indexing
        passes = 7
        for s in range(passes):
            eng.render_pass(sc.options.pass_params(s))
            g.post(fb)                                       # progressive exchange, overlapped with the next pass
            gt.post(fb)
        eng.flush()
        g.post(fb)
        gt.post(fb)
        full = g.finish()
        full_t = gt.finish()
        torch.cuda.synchronize()
        assert bool((full[..., 3] == float(passes)).all())
        assert full.cpu().numpy().tobytes() == fb.cpu().numpy().tobytes()
        assert full_t.cpu().numpy().tobytes() == fb.cpu().numpy().tobytes()
        # shards of a 3-way split: pack on each 'rank' (here three contexts on one GPU), unpack into one frame
        parts = torch.zeros_like(fb)
        for r in range(3):
            e = core.create_engine(device_id=0, rank=r, world=3, tile_size=32, stream=stream)
            sc.apply(e, lut=golden["multiscatter_lut"])
            for s in range(passes):
                e.render_pass(sc.options.pass_params(s))
            e.flush()
            packed = torch.empty((e.packed_slots(r, 3), 4), dtype=torch.float32, device=dev)
            e.pack_owned(packed.data_ptr())
            eng.unpack(r, 3, packed.data_ptr(), parts.data_ptr())
        torch.cuda.synchronize()
        assert parts.cpu().numpy().tobytes() == fb.cpu().numpy().tobytes()
    finally:
        dist.destroy_process_group()
