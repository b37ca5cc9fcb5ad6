"""rnd_render_noalloc (include/rnd.h): the frame of rnd_render with every intermediate stream in the library's
static device scratch array and the image in the caller's buffer.  Through the C ABI, two calls in a row — the
second with a smaller input — each against the existing entry point's output, bit for bit (same kernels), and
the first also against the oracle."""
import ctypes as C

import numpy as np
import pytest

from renderer_amd import VisualServer
from renderer_amd._capi import RndError
from renderer_amd.scenes import c2_grid_1m
from renderer_amd.types import Color, Light, from_rotation_x, from_translation
from tests.helpers import oracle_server
from tests.test_gpu_parity import soup_scene

pytestmark = pytest.mark.gpu


def planes(vs):
    return vs.read_triangle_ids(), vs.read_depth().view(np.uint32)


def test_noalloc_frames_equal_the_allocating_entry_point():
    import torch
    big = c2_grid_1m(width=640, height=360, scale=0.3)             # 90 000 triangles
    small = soup_scene(5, 640, 360, n_meshes=3, n_tri=60, lights=3)
    small.lights = [l for l in small.lights if l[2].kind_id() == 1]
    out = torch.full((360, 640, 4), 7, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    na = VisualServer(640, 360, device=0)
    for k, scene in enumerate((big, small)):
        ref = VisualServer(640, 360, device=0)
        scene.populate(ref)
        ref.render()
        want_rgba, (want_ids, want_depth) = ref.read_color_rgba8(), planes(ref)
        na.reset_scene()
        scene.populate(na)
        na.render_noalloc(out.data_ptr(), out.numel())
        na.finish()
        got_ids, got_depth = planes(na)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert (got_ids == want_ids).all() and (got_depth == want_depth).all(), k
        assert (got == want_rgba).all(), k
        assert (na.read_color_rgba8() == want_rgba).all(), k
        assert na.frame_stats()["n_triangles"] == scene.n_triangles
        if k == 0:
            cpu = oracle_server(640, 360)
            scene.populate(cpu)
            cpu.render()
            assert (got_ids == cpu.read_triangle_ids()).all() and (got_depth == cpu.read_depth().view(np.uint32)).all()
            assert np.abs(got.astype(int) - cpu.read_color_rgba8().astype(int)).max() <= 1
    # the allocating entry point still works on the same context afterwards, and alternates with the other
    na.render()
    assert (na.read_color_rgba8() == want_rgba).all()
    na.render_noalloc(out.data_ptr(), out.numel())
    na.render()
    na.render_noalloc(out.data_ptr(), out.numel())
    na.finish()
    torch.cuda.synchronize()
    assert (out.cpu().numpy() == want_rgba).all()


def test_noalloc_rejects_what_it_cannot_hold():
    import torch
    out = torch.zeros((360, 640, 4), dtype=torch.uint8, device="cuda")
    vs = VisualServer(640, 360, device=0)
    s = soup_scene(6, 640, 360, n_meshes=2, n_tri=30, lights=1)       # has a directional light
    s.populate(vs)
    with pytest.raises(RndError) as e:
        vs.render_noalloc(out.data_ptr(), out.numel())
    assert e.value.code == -5 and "directional" in str(e.value)
    vs.reset_scene()
    s.lights = [l for l in s.lights if l[2].kind_id() == 1]
    s.populate(vs)
    with pytest.raises(RndError) as e:
        vs.render_noalloc(out.data_ptr(), out.numel() - 4)
    assert e.value.code == -1 and "too small" in str(e.value)
    with pytest.raises(RndError):
        vs.render_noalloc(0, out.numel())
    # a frame whose streams exceed the array: 2.2 M triangles
    huge = c2_grid_1m(width=640, height=360, scale=1.5)
    vs.reset_scene()
    huge.populate(vs)
    with pytest.raises(RndError) as e:
        vs.render_noalloc(out.data_ptr(), out.numel())
    assert e.value.code == -1 and "scratch" in str(e.value)
    # and the context is still good
    vs.render()
    assert vs.read_coverage().mean() > 0.5


def test_noalloc_refusal_changes_nothing_and_the_array_has_one_holder():
    """a refused frame leaves the context as it was (the read-backs still describe the frame before), and
    the one scratch array of a GPU has one holder: when another context takes it, the previous holder's frame is gone
    (RND_ERR_NO_FRAME from its read-backs) instead of showing the other context's data."""
    import torch
    out = torch.zeros((360, 640, 4), dtype=torch.uint8, device="cuda")
    a = VisualServer(640, 360, device=0)
    s = soup_scene(8, 640, 360, n_meshes=3, n_tri=50, lights=2)
    s.lights = [l for l in s.lights if l[2].kind_id() == 1]
    s.populate(a)
    a.render_noalloc(out.data_ptr(), out.numel())
    a.finish()
    ids0, depth0 = planes(a)
    stats0 = a.frame_stats()
    # refused after the frame was laid out (2.2 M triangles do not fit the array): nothing of it may stick
    huge = c2_grid_1m(width=640, height=360, scale=1.5)
    handles = [a.asset_server.add(m) for m in huge.meshes]
    for nid, t, m in huge.instances:
        a.set_mesh_instance(1000 + nid, t, handles[m])
    with pytest.raises(RndError):
        a.render_noalloc(out.data_ptr(), out.numel())
    ids1, depth1 = planes(a)
    assert (ids1 == ids0).all() and (depth1 == depth0).all()
    assert a.frame_stats() == stats0
    # another context takes the array: a's frame is gone, b's is right
    b = VisualServer(640, 360, device=0)
    s.populate(b)
    out_b = torch.zeros_like(out)
    b.render_noalloc(out_b.data_ptr(), out_b.numel())
    b.finish()
    assert (planes(b)[0] == ids0).all()
    with pytest.raises(RndError) as e:
        a.read_depth()
    assert e.value.code == -6
    # and a renders again through the allocating entry point as if nothing had happened
    a.reset_scene()
    s.populate(a)
    a.render()
    assert (planes(a)[0] == ids0).all()
