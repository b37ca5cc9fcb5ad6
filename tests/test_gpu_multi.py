"""Multi-GPU assembly paths exercised on ONE GPU: several contexts stand in for the ranks, torch
tensors stand in for the RCCL buffers (the collective itself is covered by the gloo test)."""
import ctypes as C

import numpy as np
import pytest

from renderer_amd import VisualServer
from renderer_amd.scenes import SceneDesc, c3_grid_10m
from tests.helpers import oracle_server

pytestmark = pytest.mark.gpu


def test_sort_first_tiles_into_external_buffers_and_detile():
    import torch
    s = c3_grid_10m(width=416, height=300, scale=0.05)
    world = 3
    cpu = oracle_server(s.width, s.height)
    s.populate(cpu)
    cpu.render()
    ranks, bufs = [], []
    for r in range(world):
        vs = VisualServer(s.width, s.height, device=0)
        vs.set_tile_partition(r, world)
        info = vs.tile_info()
        buf = torch.zeros(info["n_owned"] * info["tile_px"] ** 2 * 4 + 4096, dtype=torch.uint8, device="cuda")
        vs.set_tiles_output(buf.data_ptr(), buf.numel())
        s.populate(vs)
        vs.render()
        vs.finish()
        ptr, nbytes = vs.tiles_device_ptr()
        assert ptr == buf.data_ptr() and nbytes == info["n_owned"] * info["tile_px"] ** 2 * 4
        ranks.append(vs)
        bufs.append(buf)
    torch.cuda.synchronize()
    root = ranks[0]
    for r in range(world):
        root.detile(r, world, bufs[r].data_ptr())
    root.finish()
    ptr, nbytes = root.color_device_ptr()
    frame = np.empty((s.height, s.width, 4), np.uint8)
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    assert hip.hipDeviceSynchronize() == 0
    assert hip.hipMemcpy(ctypes.c_void_p(frame.ctypes.data), ctypes.c_void_p(ptr), ctypes.c_size_t(nbytes), 2) == 0
    diff = np.abs(frame.astype(int) - cpu.read_color_rgba8().astype(int))
    assert diff.max() <= 1
    assert frame[..., 3].min() == 255          # every pixel was written by exactly the owning rank
    # the same from one [world, stride] tensor in one launch (rnd_detile_gathered), for several world sizes
    for world2 in (2, 3, 5, 8):
        stride = 0
        parts = []
        for r in range(world2):
            vs = VisualServer(s.width, s.height, device=0)
            vs.set_tile_partition(r, world2)
            s.populate(vs)
            vs.render()
            vs.finish()
            ptr, nbytes = vs.tiles_device_ptr()
            parts.append((vs, ptr, nbytes))
            stride = max(stride, nbytes)
        allbuf = torch.zeros((world2, stride), dtype=torch.uint8, device="cuda")
        for r, (vs, ptr, nbytes) in enumerate(parts):
            assert hip.hipMemcpy(ctypes.c_void_p(allbuf[r].data_ptr()), ctypes.c_void_p(ptr), ctypes.c_size_t(nbytes), 3) == 0
        root2 = parts[0][0]
        root2.detile_gathered(world2, allbuf.data_ptr(), allbuf.stride(0))
        root2.finish()
        ptr, nbytes = root2.color_device_ptr()
        frame2 = np.empty((s.height, s.width, 4), np.uint8)
        assert hip.hipDeviceSynchronize() == 0
        assert hip.hipMemcpy(ctypes.c_void_p(frame2.ctypes.data), ctypes.c_void_p(ptr), ctypes.c_size_t(nbytes), 2) == 0
        assert (frame2 == frame).all(), world2


def test_sort_last_packed_min_composite():
    import torch
    s = c3_grid_10m(width=320, height=200, scale=0.05)
    world = 2
    cpu = oracle_server(s.width, s.height)
    s.populate(cpu)
    cpu.render()
    packed, ctxs = [], []
    for r in range(world):
        vs = VisualServer(s.width, s.height, device=0)
        shard = SceneDesc(s.name, s.width, s.height, s.camera, s.camera_transform, s.meshes, s.instances[r::world], s.lights)
        shard.populate(vs)
        vs.render()
        ptr, nbytes = vs.packed_device_ptr()
        t = torch.empty(s.width * s.height, dtype=torch.int64, device="cuda")
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        assert hip.hipMemcpy(ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(ptr), ctypes.c_size_t(nbytes), 3) == 0
        packed.append(t)
        ctxs.append(vs)
    comp = torch.minimum(packed[0], packed[1])      # what ReduceOp.MIN computes across ranks
    torch.cuda.synchronize()
    ctxs[0].unpack(comp.data_ptr())
    img = ctxs[0].read_color_rgba8()
    diff = np.abs(img.astype(int) - cpu.read_color_rgba8().astype(int))
    assert diff.max() <= 1
    depth = (comp.cpu().numpy().view(np.uint64) >> np.uint64(32)).astype(np.uint32).view(np.float32).reshape(s.height, s.width)
    assert (depth == cpu.read_depth()).all()


def test_overlapped_frames_keep_their_own_output():
    """Frames in flight overlap (geometry of N+1 beside the shading of N).  As rank 0 of 2, each of a run
    of different frames goes to its own external tile buffer without any wait in between; afterwards every
    buffer must hold exactly what the same frame gives when rendered alone."""
    import torch
    from renderer_amd.types import from_translation, from_rotation_y
    s = c3_grid_10m(width=640, height=384, scale=0.12)
    vs = VisualServer(s.width, s.height, device=0)
    vs.set_tile_partition(0, 2)
    info = vs.tile_info()
    nbytes = info["n_owned"] * info["tile_px"] ** 2 * 4
    s.populate(vs)
    cams = [from_translation((0.6 * (k % 3) - 0.6, 0.3 * (k % 2), -0.5 * (k % 4))) @ from_rotation_y(0.04 * k) for k in range(10)]
    bufs = [torch.zeros(nbytes, dtype=torch.uint8, device="cuda") for _ in cams]
    torch.cuda.synchronize()
    for c, b in zip(cams, bufs):
        vs.set_camera(c, s.camera)
        vs.set_tiles_output(b.data_ptr(), b.numel())
        vs.render()
    vs.finish()
    torch.cuda.synchronize()
    alone = VisualServer(s.width, s.height, device=0)
    alone.set_tile_partition(0, 2)
    s.populate(alone)
    one = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    alone.set_tiles_output(one.data_ptr(), one.numel())
    distinct = set()
    for k, c in enumerate(cams):
        alone.set_camera(c, s.camera)
        alone.render()
        alone.finish()
        torch.cuda.synchronize()
        assert torch.equal(one, bufs[k]), f"frame {k} differs from the same frame rendered alone"
        distinct.add(int(one.to(torch.int64).sum().item()))
    assert len(distinct) >= 6          # the frames really differ from each other


@pytest.mark.parametrize("env", [{"RND_OVERLAP": "0"}, {"RND_OVERLAP": "0", "RND_EARLY_DEPTH": "1"}])
def test_single_stream_mode(env):
    """The launch mode chosen by an environment variable at context creation (everything on the caller's stream, no
    overlap of a frame's geometry with the frame before) renders the same frames as the default mode: a child process
    renders a run of frames in that mode and checks them against the oracle."""
    import os
    import subprocess
    import sys
    code = r'''
import numpy as np
from renderer_amd import VisualServer
from renderer_amd.types import from_translation
from tests.helpers import oracle_server
from tests.test_gpu_parity import soup_scene
s = soup_scene(17, 224, 160, n_meshes=3, n_tri=80, lights=3)
s.lights = [l for l in s.lights if l[2].kind_id() == 1]
gpu, cpu = VisualServer(s.width, s.height, device=0), oracle_server(s.width, s.height)
s.populate(gpu); s.populate(cpu)
for k in range(12):
    cam = from_translation((0.1 * (k % 3), 0.0, -1.0 - 0.05 * k))
    gpu.set_camera(cam, s.camera)
    gpu.render()
cpu.set_camera(cam, s.camera)
cpu.render()
assert (gpu.read_triangle_ids() == cpu.read_triangle_ids()).all()
assert (gpu.read_depth().view(np.uint32) == cpu.read_depth().view(np.uint32)).all()
assert np.abs(gpu.read_color_rgba8().astype(int) - cpu.read_color_rgba8().astype(int)).max() <= 1
print("mode ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env={**os.environ, **env, "PYTHONPATH": root},
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "mode ok" in out.stdout, (out.stdout[-500:], out.stderr[-1500:])
