"""The N > 1 assembly paths on CPU: two processes over gloo, every rank rendering with the oracle.

sort-first: ranks own stripes of tiles, rank 0 gathers the tile-major buffers (TileGather, the same
code bench.py runs over RCCL) and de-tiles them; the result must equal the unpartitioned frame.
sort-last: ranks render disjoint instance shards at full resolution; the MIN-reduced packed frame
must equal the frame of the whole scene.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, WORLD = 200, 168, 2


def _scene():
    from renderer_amd.scenes import c3_grid_10m
    return c3_grid_10m(width=W, height=H, scale=0.03)     # 16 meshes, 32 lights, ~9k triangles


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from renderer_amd.sort_first import (TileGather, depth_composite, detile_numpy, owned_tiles, pack_numpy,
                                             unpack_numpy)
        from tests.helpers import oracle_server
        scene = _scene()
        # ---- sort-first ----
        vs = oracle_server(W, H, threads=1)
        vs.set_tile_partition(rank, world)
        scene.populate(vs)
        vs.render()
        info = vs.tile_info()
        g = TileGather(rank, world, info["n_owned"], info["tile_px"], "cpu")
        buf = np.zeros(info["n_owned"] * info["tile_px"] ** 2 * 4, dtype=np.uint8)
        assert vs.api.read_tiles(vs._ctx, buf.ctypes.data) == 0
        g.tiles[:buf.size] = torch.from_numpy(buf)
        gathered = g.gather()
        if rank == 0:
            tx, ty = (W + 31) // 32, (H + 31) // 32
            frame = np.zeros((H, W, 4), np.uint8)
            for r in range(world):
                detile_numpy(frame, gathered[r].numpy()[:g.counts[r] * g.tile_bytes], owned_tiles(r, world, tx, ty),
                             info["tile_px"], tx)
            np.save(os.path.join(out_dir, "sort_first.npy"), frame)
        # ---- sort-last ----
        vs2 = oracle_server(W, H, threads=1)
        shard = type(scene)(scene.name, W, H, scene.camera, scene.camera_transform, scene.meshes,
                            scene.instances[rank::world], scene.lights)
        shard.populate(vs2)
        vs2.render()
        packed = torch.from_numpy(pack_numpy(vs2.read_depth(), vs2.read_color_rgba8()).copy())
        depth_composite(packed)
        if rank == 0:
            d, c = unpack_numpy(packed.numpy())
            np.save(os.path.join(out_dir, "sort_last_rgba.npy"), c)
            np.save(os.path.join(out_dir, "sort_last_depth.npy"), d)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_sort_first_and_sort_last(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(WORLD, port, str(tmp_path)), nprocs=WORLD, join=True)
    sys.path.insert(0, ROOT)
    from tests.helpers import oracle_server
    scene = _scene()
    ref = oracle_server(W, H, threads=1)
    scene.populate(ref)
    ref.render()
    want, want_depth = ref.read_color_rgba8(), ref.read_depth()
    assert ref.read_coverage().sum() > W * H // 2
    assert (np.load(tmp_path / "sort_first.npy") == want).all()
    assert (np.load(tmp_path / "sort_last_depth.npy") == want_depth).all()
    assert (np.load(tmp_path / "sort_last_rgba.npy") == want).all()
