"""The N > 1 path on CPU: world_size-2 (and 3) gloo process groups.  Each rank renders its tile shard (here with the CPU
oracle standing in for the GPU, which is what the tests may use it for) and the frame is put back together with
the collectives of heatray_amd.tiles; the result must equal the single-process frame bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, width, height, passes, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import oracle_lib
    from heatray_amd import scenes, tiles
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = scenes.multi_material(width, height, bounces=4)
        eng = oracle_lib.engine(rank=rank, world=world, tile_size=32)
        oracle_lib.load().ora_set_threads(eng._ctx, 2)
        sc.apply(eng, lut=np.zeros((128, 128), np.float32))
        for s in range(passes):
            eng.render_pass(sc.options.pass_params(s))
        frame = torch.from_numpy(eng.readback())
        # the shard touches exactly the pixels the ownership map gives it
        own = torch.from_numpy(tiles.owner_map(width, height, world) == rank)
        assert bool(((frame[..., 3] > 0) == own).all())
        reduced = tiles.reduce_frame(frame, dst=0)
        gathered = tiles.gather_frame(frame, rank, world, dst=0)
        # the progressive-display exchange bench.py uses for N > 1: posted after every pass, staging buffers reused
        g = tiles.FrameGatherer(width, height, rank, world, "cpu", n_buffers=2)
        for k in range(3):
            g.post(frame * float(k + 1))
        posted = g.finish()
        # the same exchange through the engine's own pack / unpack entry points (here the oracle's, on host memory)
        ge = tiles.FrameGatherer(width, height, rank, world, "cpu", n_buffers=2, engine=eng)
        ge.post(frame)
        ge.post(frame)
        via_engine = ge.finish()
        if rank == 0:
            assert via_engine.numpy().tobytes() == gathered.numpy().tobytes()
        assert bool((frame[..., 3][~own] == 0).all())  # the local accumulator is untouched by the collectives
        if rank == 0:
            q.put((reduced.numpy(), gathered.numpy(), posted.numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_shards_reduce_and_gather_to_the_full_frame(world):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from heatray_amd import scenes
    width, height, passes = 100, 70, 2   # not a multiple of the tile size: edge tiles are partial
    sc = scenes.multi_material(width, height, bounces=4)
    eng = oracle_lib.engine()
    sc.apply(eng, lut=np.zeros((128, 128), np.float32))
    for s in range(passes):
        eng.render_pass(sc.options.pass_params(s))
    full = eng.readback()

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, width, height, passes, q)) for r in range(world)]
    for p in procs:
        p.start()
    reduced, gathered, posted = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert reduced.tobytes() == full.tobytes()
    assert gathered.tobytes() == full.tobytes()
    assert posted.tobytes() == (full * 3.0).tobytes()


def test_ownership_map_matches_the_engines():
    from heatray_amd import tiles
    m = tiles.owner_map(100, 70, 3)
    assert m.shape == (70, 100) and set(np.unique(m)) == {0, 1, 2}
    tx, ty = tiles.tile_grid(100, 70)
    assert (tx, ty) == (4, 3)
    assert list(tiles.owned_tiles(100, 70, 1, 3)) == [1, 4, 7, 10]
    assert m[0, 0] == 0 and m[0, 32] == 1 and m[32, 0] == (4 % 3)
