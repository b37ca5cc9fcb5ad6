"""The limits stated in include/rnd.h beside rnd_create / rnd_set_light / rnd_set_render_size: 256 lights per context and
8192 x 8192 pixels.  At the limit the frame is rendered and agrees with the oracle; one beyond it the call is refused
with RND_ERR_INVALID and the context stays usable."""
import numpy as np
import pytest

from renderer_amd import VisualServer
from renderer_amd.types import Color, Light, from_translation
from renderer_amd._capi import RndError
from tests.helpers import oracle_server
from tests.test_gpu_parity import assert_parity, soup_scene

pytestmark = pytest.mark.gpu
RND_ERR_INVALID = -1


def many_lights(n, seed=5):
    r = np.random.default_rng(seed)
    out = []
    for k in range(n):
        p = r.uniform((-2.5, -2.0, 0.5), (2.5, 2.0, 6.0))
        out.append((1000 + k, from_translation(p), Light.point(float(r.uniform(0.8, 3.0))).with_color(Color(*r.uniform(0.1, 1.0, 3), 1.5))))
    return out


def test_256_lights_render_and_the_257th_is_refused():
    s = soup_scene(31, 224, 160, n_meshes=3, n_tri=150, size=0.5, lights=0)
    s.lights = many_lights(256)
    gpu, cpu = VisualServer(s.width, s.height, device=0), oracle_server(s.width, s.height)
    for vs in (gpu, cpu):
        s.populate(vs)
        vs.render()
    assert gpu.frame_stats()["n_lights"] == 256
    assert_parity(gpu, cpu, min_covered=2000)
    extra = many_lights(1, seed=9)[0]
    with pytest.raises(RndError) as e:
        gpu.set_light(5000, extra[1], extra[2])
    assert e.value.code == RND_ERR_INVALID and "256" in str(e.value)
    # re-setting a light that exists is not a 257th, and the context still renders the same frame
    gpu.set_light(s.lights[0][0], s.lights[0][1], s.lights[0][2])
    gpu.render()
    assert_parity(gpu, cpu, min_covered=2000)
    gpu.close(); cpu.close()


def test_largest_render_size_and_one_beyond():
    # 8192 x 64: the widest frame (tile coordinates are packed in 8 bits: 256 tiles of 32 pixels); a soup stretched over it
    w, h = 8192, 64
    s = soup_scene(33, w, h, n_meshes=2, n_tri=200, size=0.8, lights=2, spread=(300.0, 0.5), depth=(2.0, 4.0))      # (128 : 1: the view is ~600 units wide at these depths)
    s.lights = [l for l in s.lights if l[2].kind_id() == 1]
    gpu, cpu = VisualServer(w, h, device=0), oracle_server(w, h)
    for vs in (gpu, cpu):
        s.populate(vs)
        vs.render()
    assert_parity(gpu, cpu, min_covered=500)
    ids = gpu.read_triangle_ids()
    assert (ids[:, w - 64:] != 0xFFFFFFFF).any() or (ids[:, :64] != 0xFFFFFFFF).any()      # something reaches the far tile columns
    gpu.close(); cpu.close()
    # ... and the tallest
    s2 = soup_scene(34, 64, 8192, n_meshes=2, n_tri=200, size=0.3, lights=1, spread=(1.0, 2.0), depth=(2.0, 4.0))       # (1 : 128: the view is a few hundredths of a unit wide, every triangle that crosses it spans it)
    s2.lights = [l for l in s2.lights if l[2].kind_id() == 1]
    gpu, cpu = VisualServer(64, 8192, device=0), oracle_server(64, 8192)
    for vs in (gpu, cpu):
        s2.populate(vs)
        vs.render()
    assert_parity(gpu, cpu, min_covered=500)
    gpu.close(); cpu.close()
    for bad in ((8193, 64), (64, 8193), (0, 10)):
        with pytest.raises(RndError) as e:
            VisualServer(*bad, device=0)
        assert e.value.code == RND_ERR_INVALID
    vs = VisualServer(64, 64, device=0)
    with pytest.raises(RndError) as e:
        vs.set_render_size((8193, 64))
    assert e.value.code == RND_ERR_INVALID
    vs.set_render_size((128, 96))          # the refused call left the context usable
    s3 = soup_scene(35, 128, 96, n_meshes=1, n_tri=50, lights=1)
    s3.lights = [l for l in s3.lights if l[2].kind_id() == 1]
    s3.populate(vs)
    vs.render()
    assert (vs.read_triangle_ids() != 0xFFFFFFFF).any()
    vs.close()
