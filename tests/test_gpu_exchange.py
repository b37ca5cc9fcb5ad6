"""The multi-GPU exchange paths with the HIP library as the renderer: the local composite kernel, the C-level
RCCL entry points on a one-rank communicator, and — two processes over gloo sharing this box's GPU — the
sort-first gather and the sort-last direct-send composite as bench.py drives them (TileGather, StripComposite).
Every assembled frame is compared with the single-context frame and with the oracle's."""
import os
import socket
import sys

import numpy as np
import pytest

from renderer_amd import VisualServer
from renderer_amd.scenes import SceneDesc
from tests.helpers import oracle_server
from tests.test_gpu_more_paths import c5_small

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_min_packed_against_torch():
    import torch
    vs = VisualServer(64, 64, device=0)
    g = torch.Generator(device="cpu").manual_seed(3)
    for n, count, stride in ((1, 10, 10), (3, 1001, 1002), (8, 4096, 4096), (5, 7, 64)):
        strips = torch.randint(0, 2 ** 62, (n, stride), generator=g, dtype=torch.int64).cuda()
        out = torch.full((stride,), -1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        vs.min_packed(strips.data_ptr(), n, stride, out.data_ptr(), count)
        vs.finish()
        torch.cuda.synchronize()
        assert torch.equal(out[:count], strips[:, :count].amin(dim=0)) and (out[count:] == -1).all()
    with pytest.raises(Exception):
        vs.min_packed(strips.data_ptr(), 2, 63, out.data_ptr(), 7)          # odd stride


def test_c_level_exchanges_on_a_one_rank_communicator():
    """rnd_comm_make_id / rnd_comm_init load RCCL and build a communicator; with one rank the gather has nothing to
    move and the composite is copy -> minimum of one strip -> unpack: the frame must come out as rendered."""
    s = c5_small(352, 200)
    vs = VisualServer(s.width, s.height, device=0)
    s.populate(vs)
    vs.render()
    want = vs.read_color_rgba8()
    vs.comm_init(vs.comm_make_id(), 0, 1)
    vs.packed_device_ptr()                      # turns the packed output on (and renders once)
    vs.render()
    vs.composite_packed()
    assert (vs.read_color_rgba8() == want).all()
    vs.gather_tiles()
    vs.comm_destroy()
    with pytest.raises(Exception):
        vs.composite_packed()                   # no communicator any more


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from renderer_amd.sort_first import StripComposite, TileGather
        torch.cuda.set_device(0)
        s = c5_small(416, 300)
        stream = torch.cuda.Stream()
        # ---- sort-first: stripes of tiles, gather, one de-tile launch on rank 0 ----
        vs = VisualServer(s.width, s.height, device=0)
        vs.set_stream(stream.cuda_stream)
        vs.set_tile_partition(rank, world)
        info = vs.tile_info()
        g = TileGather(rank, world, info["n_owned"], info["tile_px"], "cuda")
        vs.set_tiles_output(g.tiles.data_ptr(), g.tiles.numel())
        s.populate(vs)
        with torch.cuda.stream(stream):
            for _ in range(3):                  # pipelined frames, as the bench
                vs.render()
            vs.finish()
            g.gather()
            if rank == 0:
                vs.detile_gathered(world, g.gathered_all.data_ptr(), g.gathered_all.stride(0))
        if rank == 0:
            # (rnd_read_color_rgba8 on a partitioned context shows that context's own tiles; the assembled frame is
            # what rnd_color_device_ptr points at)
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")
            stream.synchronize()
            ptr, nbytes = vs.color_device_ptr()
            frame = np.empty((s.height, s.width, 4), np.uint8)
            assert hip.hipMemcpy(ctypes.c_void_p(frame.ctypes.data), ctypes.c_void_p(ptr), ctypes.c_size_t(nbytes), 2) == 0
            np.save(os.path.join(out_dir, "sort_first.npy"), frame)
        # ---- sort-last: instance shards, strips all-to-all, minimum, strips to rank 0 ----
        vs2 = VisualServer(s.width, s.height, device=0)
        vs2.set_stream(stream.cuda_stream)
        shard = SceneDesc(s.name, s.width, s.height, s.camera, s.camera_transform, s.meshes, s.instances[rank::world], s.lights)
        shard.populate(vs2)
        sc = StripComposite(rank, world, s.width * s.height, "cuda")
        with torch.cuda.stream(stream):
            for b in (0, 1, 0):                 # both buffer pairs
                vs2.set_packed_output(sc.frames[b].data_ptr(), sc.frames[b].numel() * 8)
                vs2.render()
                vs2.finish()
                sc.exchange(b)
                sc.reduce(vs2, b)
                vs2.finish()
                done = sc.collect(b)
            if rank == 0:
                vs2.unpack(done.data_ptr())
        if rank == 0:
            np.save(os.path.join(out_dir, "sort_last.npy"), vs2.read_color_rgba8())
            np.save(os.path.join(out_dir, "sort_last_packed.npy"), done.cpu().numpy().reshape(-1)[:s.width * s.height])
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_over_gloo_drive_the_hip_path(tmp_path):
    import torch.multiprocessing as mp
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    s = c5_small(416, 300)
    one = VisualServer(s.width, s.height, device=0)
    s.populate(one)
    one.render()
    want, want_depth = one.read_color_rgba8(), one.read_depth()
    cpu = oracle_server(s.width, s.height)
    s.populate(cpu)
    cpu.render()
    assert np.abs(want.astype(int) - cpu.read_color_rgba8().astype(int)).max() <= 1 and (want_depth == cpu.read_depth()).all()
    assert (np.load(tmp_path / "sort_first.npy") == want).all()
    packed = np.load(tmp_path / "sort_last_packed.npy").view(np.uint64)
    depth = (packed >> np.uint64(32)).astype(np.uint32).view(np.float32).reshape(s.height, s.width)
    assert (depth == want_depth).all()
    differs = (np.load(tmp_path / "sort_last.npy") != want).any(axis=-1)
    assert differs.mean() <= 1e-4          # (equal-depth ties between shards resolve by colour bits: none expected here)
