"""Multi-GPU frame assembly: one process per GPU, torch.distributed (RCCL on GPUs, gloo in CPU tests).

sort-first  every rank rasterizes/shades the screen tiles it owns (full-width stripes of 4 tile rows
            dealt round-robin, rnd_set_tile_partition); the finished tile-major RGBA8 buffers are
            gathered to rank 0 (each peer sends its own buffer straight to rank 0: over xGMI that is
            one link per peer, no ring) and scattered into the row-major frame there.
sort-last   every rank renders its own mesh shard at full resolution into a packed
            (depth bits << 32 | rgba8) frame; a MIN reduction to rank 0 is the depth composite
            (depth in [0,1] makes the IEEE bits order-preserving; the packed value is < 2^62, so the
            signed int64 view orders the same way).

The reference is single-device; this module has no counterpart there.
"""
import numpy as np
import torch
import torch.distributed as dist

OWNER_GROUP = 2     # tile rows per ownership stripe (csrc/device_types.h kOwnerGroup)


def owned_tiles(rank, world, tiles_x, tiles_y):
    """Ascending tile indices owned by `rank` (must match rnd::VisualServer::owned_tiles): stripes of OWNER_GROUP tile rows dealt
    round-robin from the LAST rank, so that rank 0, which assembles the frame, never holds the larger share."""
    return [ty * tiles_x + tx for ty in range(tiles_y) if world <= 1 or world - 1 - (ty // OWNER_GROUP) % world == rank
            for tx in range(tiles_x)]


def detile_numpy(frame, tiles_rgba8, tile_list, tile_px, tiles_x):
    """Host version of rnd_detile: scatter tile-major RGBA8 tiles into a row-major (H, W, 4) frame."""
    h, w = frame.shape[:2]
    t = np.asarray(tiles_rgba8, dtype=np.uint8).reshape(-1, tile_px, tile_px, 4)
    for k, tile in enumerate(tile_list):
        x0, y0 = (tile % tiles_x) * tile_px, (tile // tiles_x) * tile_px
        x1, y1 = min(x0 + tile_px, w), min(y0 + tile_px, h)
        frame[y0:y1, x0:x1] = t[k, :y1 - y0, :x1 - x0]
    return frame


class TileGather:
    """Equal-sized gather buffers for the tile-major outputs of all ranks (padded to the largest)."""

    def __init__(self, rank, world, n_owned, tile_px, device):
        self.rank, self.world, self.tile_bytes = rank, world, tile_px * tile_px * 4
        t = torch.tensor([n_owned], dtype=torch.int64, device="cpu" if dist.get_backend() == "gloo" else device)
        counts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(counts, t)
        self.counts = [int(c.item()) for c in counts]
        self.tiles = torch.zeros(max(max(self.counts), 1) * self.tile_bytes, dtype=torch.uint8, device=device)
        # rank 0: one contiguous [world, stride] tensor, so one kernel can scatter all ranks' tiles
        # (rnd_detile_gathered); the gather list is its rows
        self.gathered_all = torch.empty((world, self.tiles.numel()), dtype=torch.uint8, device=device) if rank == 0 else None
        self.gathered = [self.gathered_all[r] for r in range(world)] if rank == 0 else None

    def gather(self):
        """Collect every rank's `tiles` buffer on rank 0 (a send/recv group, no ring)."""
        if dist.get_backend() == "gloo" and self.tiles.is_cuda:
            # rehearsal on a box with fewer GPUs than ranks: gloo gathers host tensors only
            host = self.tiles.cpu()
            parts = [torch.empty_like(host) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(host, parts, dst=0)
            if self.rank == 0:
                for dst, src in zip(self.gathered, parts):
                    dst.copy_(src)
            return self.gathered
        dist.gather(self.tiles, self.gathered, dst=0)
        return self.gathered


def depth_composite(packed):
    """Sort-last composite: element-wise MIN of the packed int64 frames onto rank 0 (in place)."""
    dist.reduce(packed, dst=0, op=dist.ReduceOp.MIN)
    return packed


def pack_numpy(depth, rgba8):
    """Host version of the packed frame: (f32 depth bits << 32) | rgba8 little-endian word."""
    d = np.ascontiguousarray(depth, dtype=np.float32).view(np.uint32).astype(np.uint64)
    c = np.ascontiguousarray(rgba8, dtype=np.uint8).view(np.uint32).reshape(d.shape).astype(np.uint64)
    return ((d << np.uint64(32)) | c).view(np.int64)


def unpack_numpy(packed):
    p = np.asarray(packed).view(np.uint64)
    depth = (p >> np.uint64(32)).astype(np.uint32).view(np.float32)
    rgba = (p & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.uint8).reshape(p.shape + (4,))
    return depth, rgba


CLEAR_KEY = (0x3F800000 << 32) | 0xFFFFFFFF        # depth 1.0 in the high half: loses every comparison


class StripComposite:
    """Sort-last direct-send composite: the packed frame is cut into `world` equal strips, strip s
    of every rank goes to rank s (all-to-all: every pair over its own xGMI link, no ring), rank s takes the
    element-wise minimum (rnd_min_packed), the finished strips are gathered on rank 0.  Two sets of buffers, so the
    exchange of one frame can run beside the rendering of the next."""

    def __init__(self, rank, world, n_pixels, device):
        self.rank, self.world, self.n_pixels = rank, world, n_pixels
        self.strip = ((n_pixels + world - 1) // world + 1) & ~1          # words per strip, even (16-byte loads)
        n = world * self.strip
        # render targets of the packed frame (rnd_set_packed_output), padded to whole strips with the clear key
        self.frames = [torch.full((n,), CLEAR_KEY, dtype=torch.int64, device=device) for _ in range(2)]
        self.recv = [torch.empty((world, self.strip), dtype=torch.int64, device=device) for _ in range(2)]
        self.mine = [torch.empty(self.strip, dtype=torch.int64, device=device) for _ in range(2)]
        self.done = [torch.empty((world, self.strip), dtype=torch.int64, device=device) if rank == 0 else None for _ in range(2)]

    def exchange(self, b):
        """All-to-all of buffer pair b's strips (collective)."""
        frame, recv = self.frames[b], self.recv[b]
        if dist.get_backend() == "gloo":
            # rehearsal on a box with fewer GPUs than ranks: gloo has no all-to-all; every rank gets every frame on the
            # host and keeps its strip of each
            host = frame.cpu()
            parts = [torch.empty_like(host) for _ in range(self.world)]
            dist.all_gather(parts, host)
            s = self.strip
            recv.copy_(torch.stack([p[self.rank * s:(self.rank + 1) * s] for p in parts]))
            return
        dist.all_to_all_single(recv.view(-1), frame)

    def reduce(self, vs, b):
        """Minimum of the received strips on the context's stream."""
        vs.min_packed(self.recv[b].data_ptr(), self.world, self.strip, self.mine[b].data_ptr(), self.strip)

    def collect(self, b):
        """Finished strips to rank 0 (collective); returns the whole packed frame there."""
        mine, done = self.mine[b], self.done[b]
        if dist.get_backend() == "gloo":
            host = mine.cpu()
            parts = [torch.empty_like(host) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(host, parts, dst=0)
            if self.rank == 0:
                done.copy_(torch.stack(parts))
            return done
        dist.gather(mine, [done[r] for r in range(self.world)] if self.rank == 0 else None, dst=0)
        return done
