"""Pixel-tile sharding of one frame over the ranks of a node and the collective that puts the
HDR accumulation buffer back together.

One process per GPU.  Rank r of `world` renders the 32x32-pixel tiles t with t % world == r (tiles numbered row-major
from the BOTTOM-left, like the accumulation buffer) into a full-frame RGBA32F buffer that is zero elsewhere, so

  * `reduce_frame`  — `dist.reduce(SUM)` of the whole buffer (33.2 MB at 1080p) — is exact: every pixel has one
    non-zero contributor;
  * `gather_frame`  — every rank packs only the tiles it owns (1/world of the buffer) and rank 0 scatters the
    gathered tiles into place: world x fewer bytes over xGMI, bit-identical result.

Both work with any torch.distributed backend ("nccl" == RCCL on ROCm, "gloo" in the CPU tests).
"""
import numpy as np


def tile_grid(width, height, tile=32):
    return (width + tile - 1) // tile, (height + tile - 1) // tile


def owner_map(width, height, world, tile=32):
    """int32 [H, W]: rank that owns each pixel (same rule as hr_ctx_desc.rank/world in libhrcore)."""
    tx, ty = tile_grid(width, height, tile)
    t = (np.arange(height)[:, None] // tile) * tx + (np.arange(width)[None, :] // tile)
    return (t % world).astype(np.int32)


def owned_tiles(width, height, rank, world, tile=32):
    """Tile ids owned by `rank`, ascending — the order libhrcore enumerates them in."""
    tx, ty = tile_grid(width, height, tile)
    return np.arange(rank, tx * ty, world)


def _tile_slices(t, width, height, tile):
    tx, _ = tile_grid(width, height, tile)
    x0, y0 = (t % tx) * tile, (t // tx) * tile
    return slice(y0, min(y0 + tile, height)), slice(x0, min(x0 + tile, width))


def reduce_frame(frame, dst=0, out=None):
    """Sum-reduce the full-frame buffers of all ranks to `dst` (torch tensors [H, W, 4]).  `frame` itself is left
    untouched (it is this rank's accumulator); returns the reduced tensor on `dst`, the scratch copy elsewhere."""
    import torch.distributed as dist
    out = frame.clone() if out is None else out.copy_(frame)
    dist.reduce(out, dst=dst, op=dist.ReduceOp.SUM)
    return out


def pack_owned(frame, rank, world, tile=32):
    """[n_owned, tile, tile, 4] tensor with this rank's tiles (edge tiles zero-padded)."""
    import torch
    h, w = frame.shape[0], frame.shape[1]
    tiles = owned_tiles(w, h, rank, world, tile)
    packed = torch.zeros((len(tiles), tile, tile, 4), dtype=frame.dtype, device=frame.device)
    for i, t in enumerate(tiles):
        ys, xs = _tile_slices(int(t), w, h, tile)
        packed[i, : ys.stop - ys.start, : xs.stop - xs.start] = frame[ys, xs]
    return packed


def gather_frame(frame, rank, world, dst=0, tile=32):
    """All ranks send their owned tiles to `dst`, which returns the assembled full frame (others return None)."""
    import torch
    import torch.distributed as dist
    h, w = frame.shape[0], frame.shape[1]
    n_max = len(owned_tiles(w, h, 0, world, tile))  # rank 0 owns the most tiles
    packed = pack_owned(frame, rank, world, tile)
    if packed.shape[0] < n_max:  # equal-sized contributions for gather
        pad = torch.zeros((n_max - packed.shape[0],) + tuple(packed.shape[1:]), dtype=packed.dtype, device=packed.device)
        packed = torch.cat([packed, pad])
    bufs = [torch.empty_like(packed) for _ in range(world)] if rank == dst else None
    dist.gather(packed, bufs, dst=dst)
    if rank != dst:
        return None
    full = torch.zeros_like(frame)
    for r in range(world):
        for i, t in enumerate(owned_tiles(w, h, r, world, tile)):
            ys, xs = _tile_slices(int(t), w, h, tile)
            full[ys, xs] = bufs[r][i, : ys.stop - ys.start, : xs.stop - xs.start]
    return full


def owned_pixel_index(width, height, rank, world, tile=32):
    """int64 flat pixel indices (row-major over [H, W]) of the pixels `rank` owns, tile by tile in ascending tile id,
    rows bottom-up inside a tile — i.e. exactly the pixels libhrcore writes for this rank, edge tiles cropped."""
    idx = []
    for t in owned_tiles(width, height, rank, world, tile):
        ys, xs = _tile_slices(int(t), width, height, tile)
        yy = np.arange(ys.start, ys.stop)[:, None]
        xx = np.arange(xs.start, xs.stop)[None, :]
        idx.append((yy * width + xx).reshape(-1))
    return np.concatenate(idx).astype(np.int64) if idx else np.zeros((0,), np.int64)


class FrameGatherer:
    """Progressive-display exchange for a tile-sharded frame: every `post()` packs this rank's owned pixels
    (1/world of the frame) and gathers them on `dst`; on CUDA the gather runs on a side stream from one of
    `n_buffers` staging buffers, so it overlaps the next pass's kernels (xGMI is point-to-point: `dst` receives
    from each peer over its own link).  `finish()` returns the assembled frame of the last post on `dst`.

    With `engine` (an `_ffi.Engine` on the same frame) packing and unpacking are the core's own kernels
    (`hr_frame_pack_owned` / `hr_frame_unpack`: coalesced tile copies — torch's row gather with int64 indices costs
    0.44 ms for a 1080p frame on MI355X, the kernel 0.02 ms); without it plain torch indexing is used (any device).

    Bit-exact by construction: pixels are copied, never summed."""

    def __init__(self, width, height, rank, world, device, tile=32, dst=0, n_buffers=2, overlap=True, engine=None, host_staged=False):
        import torch
        self.torch = torch
        self.w, self.h, self.rank, self.world, self.dst = width, height, rank, world, dst
        self.device = torch.device(device)
        self.engine = engine
        # host_staged: the collective runs on host copies of the staging buffers (a gloo group with the frames on a GPU: the one-device
        # rehearsal of bench.py); packing and unpacking stay on the device
        self.host_staged = bool(host_staged)
        if engine is not None:
            self.idx = None
            self.n_max = max(engine.packed_slots(r, world) for r in range(world))
        else:
            self.idx = [torch.from_numpy(owned_pixel_index(width, height, r, world, tile)).to(self.device) if (r == rank or rank == dst) else None
                        for r in range(world)]
            self.n_max = max(len(owned_pixel_index(width, height, r, world, tile)) for r in range(world))
        self.send = [torch.zeros((self.n_max, 4), dtype=torch.float32, device=self.device) for _ in range(n_buffers)]
        self.recv = [[torch.empty((self.n_max, 4), dtype=torch.float32, device=self.device) for _ in range(world)] for _ in range(n_buffers)] \
            if rank == dst else None
        self.full = torch.zeros((height, width, 4), dtype=torch.float32, device=self.device) if rank == dst else None
        self.cuda = self.device.type == "cuda"
        self.side = torch.cuda.Stream(device=self.device) if (self.cuda and overlap and not self.host_staged) else None
        self.free_ev = [None] * n_buffers  # recorded on the side stream when a staging buffer may be reused
        self.turn = 0
        self.posted = False

    def _stream_handle(self, stream):
        return stream.cuda_stream if (self.cuda and stream is not None) else None

    def post(self, frame):
        import torch.distributed as dist
        torch = self.torch
        b = self.turn % len(self.send)
        self.turn += 1
        cur = torch.cuda.current_stream(self.device) if self.cuda else None
        if self.side is not None and self.free_ev[b] is not None:
            cur.wait_event(self.free_ev[b])
        if self.engine is not None:
            # the engine reads its own accumulation buffer (`frame` is that buffer); ordered on the current stream
            self.engine.pack_owned(self.send[b].data_ptr(), self._stream_handle(cur))
        else:
            own = self.idx[self.rank]
            if own.numel():
                torch.index_select(frame.view(-1, 4), 0, own, out=self.send[b][: own.numel()])
        if self.side is not None:
            ready = torch.cuda.Event()
            ready.record(cur)
            self.side.wait_event(ready)
            with torch.cuda.stream(self.side):
                self._exchange(b, dist, self.side)
                ev = torch.cuda.Event()
                ev.record(self.side)
                self.free_ev[b] = ev
        else:
            self._exchange(b, dist, cur)
        self.posted = True

    def _exchange(self, b, dist, stream):
        if self.host_staged:
            host = self.send[b].cpu()  # (synchronises with the packing kernel on the current stream)
            got = [self.torch.empty_like(host) for _ in range(self.world)] if self.rank == self.dst else None
            dist.gather(host, got, dst=self.dst)
            if self.rank == self.dst:
                for r in range(self.world):
                    self.recv[b][r].copy_(got[r])
        else:
            dist.gather(self.send[b], self.recv[b] if self.rank == self.dst else None, dst=self.dst)
        if self.rank == self.dst:
            if self.engine is not None:
                for r in range(self.world):
                    self.engine.unpack(r, self.world, self.recv[b][r].data_ptr(), self.full.data_ptr(), self._stream_handle(stream))
            else:
                flat = self.full.view(-1, 4)
                for r in range(self.world):
                    n = self.idx[r].numel()
                    if n:
                        flat.index_copy_(0, self.idx[r], self.recv[b][r][:n])

    def finish(self):
        """Join the side stream into the current one; returns the assembled frame on `dst`, None elsewhere."""
        if self.side is not None:
            self.torch.cuda.current_stream(self.device).wait_stream(self.side)
        return self.full if self.rank == self.dst else None
