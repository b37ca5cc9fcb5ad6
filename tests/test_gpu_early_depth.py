"""The early-depth variant of the setup kernel (frames with depth complexity): the same image, bit for
bit, as the plain variant and as the CPU oracle.  Near-to-far item order for the work units, the two passes around the
coarse-depth kernel (k_hiz), the per-fragment depth test in front of the visibility atomic, and the host's switch
from the frame reports (rnd_frame_stats.early_depth / n_fragments).
"""
import numpy as np
import pytest

from renderer_amd import VisualServer
from renderer_amd.scenes import c4_layers_50m
from tests.helpers import oracle_server
from tests.test_gpu_parity import assert_parity, soup_scene

pytestmark = pytest.mark.gpu


def layers(scale=0.2, w=640, h=360):
    # ten stacked height fields, drawn in shuffled order: 2 M sub-pixel triangles, ~13 bbox pixels per pixel
    return c4_layers_50m(width=w, height=h, scale=scale)


def planes(vs):
    return vs.read_triangle_ids(), vs.read_depth().view(np.uint32), vs.read_color_rgba8()


@pytest.mark.parametrize("clip", ["1", "0"], ids=["clip-stored", "clip-rederived"])
def test_forced_early_depth_matches_plain_and_oracle(monkeypatch, clip):
    monkeypatch.setenv("RND_STORE_CLIP", clip)
    scene = layers()
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("RND_EARLY_DEPTH", mode)
        gpu = VisualServer(scene.width, scene.height, device=0)
        scene.populate(gpu)
        gpu.render()
        st = gpu.frame_stats()
        assert st["early_depth"] == int(mode)
        if mode == "0":
            assert st["n_fragments"] > 6 * scene.width * scene.height
        out[mode] = planes(gpu) + (st["n_fragments"],)
        if mode == "1":
            cpu = oracle_server(scene.width, scene.height)
            scene.populate(cpu)
            cpu.render()
            assert_parity(gpu, cpu, min_covered=scene.width * scene.height // 2)
            cpu.close()
        gpu.close()
    for a, b in zip(out["0"][:3], out["1"][:3]):
        assert (a == b).all()
    # the complexity measure: exact in the plain variant (bboxes of up to 32 pixel centres); the early-depth variant
    # rasterizes bboxes of up to 64 at setup and counts the triangles of the units it turns away whole at the rate of the
    # last plain frame it saw (none here: forced on from the first frame)
    assert out["1"][3] > 0


def test_switches_on_by_itself_and_off_again(monkeypatch):
    monkeypatch.delenv("RND_EARLY_DEPTH", raising=False)
    deep, flat = layers(), layers(scale=0.02)          # 2 M and 20 k triangles behind the same camera
    gpu = VisualServer(deep.width, deep.height, device=0)
    cpu = oracle_server(deep.width, deep.height)
    deep.populate(cpu)
    cpu.render()
    deep.populate(gpu)
    seen = []
    for _ in range(6):                                   # pipelined frames, no read-back: the switch comes from the reports
        gpu.render()
        gpu.finish()
        seen.append(gpu.frame_stats()["early_depth"])
    assert seen[0] == 0 and seen[-1] == 1, seen
    assert_parity(gpu, cpu, min_covered=deep.width * deep.height // 2)
    # ... and stays on: the units its second pass turns away whole keep counting in the measure
    for _ in range(12):
        gpu.render()
        gpu.finish()
        assert gpu.frame_stats()["early_depth"] == 1
    assert gpu.frame_stats()["n_fragments"] > 6 * deep.width * deep.height
    # a frame without depth complexity turns it off again
    gpu.reset_scene()
    flat.populate(gpu)
    seen = []
    for _ in range(6):
        gpu.render()
        gpu.finish()
        seen.append(gpu.frame_stats()["early_depth"])
    assert seen[-1] == 0, seen
    cpu.reset_scene()
    flat.populate(cpu)
    cpu.render()
    assert_parity(gpu, cpu, min_covered=1000)
    gpu.close(); cpu.close()


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_forced_early_depth_on_soups_with_big_and_clipped_triangles(monkeypatch, seed):
    # triangles of every size: records, bins and the clipper beside the two direct passes; ties between equal depths
    monkeypatch.setenv("RND_EARLY_DEPTH", "1")
    s = soup_scene(seed, 233, 171, n_meshes=6, n_tri=400, size=0.05 if seed != 13 else 0.6, lights=2)
    s.lights = [l for l in s.lights if l[2].kind_id() == 1]
    gpu = VisualServer(s.width, s.height, device=0)
    cpu = oracle_server(s.width, s.height)
    for vs in (gpu, cpu):
        s.populate(vs)
        vs.render()
    assert gpu.frame_stats()["early_depth"] == 1
    assert_parity(gpu, cpu, min_covered=500)
    gpu.close(); cpu.close()


def test_forced_early_depth_on_a_tile_partition(monkeypatch):
    monkeypatch.setenv("RND_EARLY_DEPTH", "1")
    scene = layers(scale=0.1, w=512, h=400)
    whole = VisualServer(scene.width, scene.height, device=0)
    scene.populate(whole)
    whole.render()
    ref = whole.read_triangle_ids()
    got = np.full_like(ref, 0xFFFFFFFF)
    for rank in range(3):
        part = VisualServer(scene.width, scene.height, device=0)
        part.set_tile_partition(rank, 3)
        scene.populate(part)
        part.render()
        ids = part.read_triangle_ids()
        own = ids != 0xFFFFFFFF
        assert not (own & (got != 0xFFFFFFFF)).any()
        got[own] = ids[own]
        part.close()
    assert (got == ref).all()
    whole.close()


@pytest.mark.parametrize("seed", [21, 22, 23, 24, 25, 26])
def test_units_turned_away_whole_random_occluders(monkeypatch, seed):
    """The unit-level rejection of the second early-depth pass on random content: a finely tessellated wall somewhere in
    the middle of the depth range (it goes the direct path, so k_hiz holds real depths; it covers only part of the
    screen), and thirty small meshes in front of it, behind it, through it and beside it, some rotated so that their
    boxes are far from axis-aligned.  Whatever is turned away must not have been visible: ids, depth bits and colours
    against the oracle.  The odd seeds must have units that were turned away (rnd_frame_stats.n_units_hidden)."""
    from renderer_amd.scenes import SceneDesc, heightfield_mesh
    from renderer_amd.types import Camera, from_translation, from_rotation_x, from_rotation_y
    r = np.random.default_rng(seed)
    w, h = int(r.integers(200, 420)), int(r.integers(150, 300))
    s = SceneDesc(f"occl{seed}", w, h, Camera(aspect_ratio=w / h), from_translation((0.0, 0.0, -1.0)))
    # the wall: nearest item for most of its pixels
    x0, x1 = sorted(r.uniform(-3.0, 3.0, 2)); y0, y1 = sorted(r.uniform(-2.0, 2.0, 2))
    # (odd seeds: the wall is the nearest item, so it alone is the first pass and much of the rest lies behind it)
    wall_z = float(r.uniform(1.0, 1.3)) if seed % 2 else float(r.uniform(2.0, 3.5))
    s.meshes.append(heightfield_mesh(160, 120, x0 - 0.5, x1 + 0.5, y0 - 0.5, y1 + 0.5, wall_z, 0.15, seed))
    s.instances.append((1, from_rotation_y(float(r.uniform(-0.2, 0.2))), 0))
    for m in range(30):
        nx, ny = int(r.integers(8, 40)), int(r.integers(8, 40))
        cx, cy = r.uniform(-3.5, 3.5), r.uniform(-2.5, 2.5)
        sz = float(r.uniform(0.2, 1.2))
        s.meshes.append(heightfield_mesh(nx, ny, -sz, sz, -sz, sz, 0.0, 0.1, seed + m))
        t = from_translation((cx, cy, float(r.uniform(1.6 if seed % 2 else 1.0, 7.0)))) @ from_rotation_y(float(r.uniform(-1.0, 1.0))) @ from_rotation_x(float(r.uniform(-1.0, 1.0)))
        s.instances.append((2 + m, t, 1 + m))
    from renderer_amd.types import Color, Light
    s.lights.append((100, from_translation((0.0, 0.0, 0.5)), Light.point(12.0).with_color(Color(1.0, 0.9, 0.8, 2.0))))
    counts = {}
    cpu = oracle_server(w, h)
    s.populate(cpu)
    cpu.render()
    for mode in ("0", "1"):
        monkeypatch.setenv("RND_EARLY_DEPTH", mode)
        gpu = VisualServer(w, h, device=0)
        s.populate(gpu)
        gpu.render()
        st = gpu.frame_stats()
        assert st["early_depth"] == int(mode)
        counts[mode] = st["n_units_hidden"]
        assert_parity(gpu, cpu, min_covered=w * h // 8)
        gpu.close()
    cpu.close()
    assert counts["0"] == 0
    if seed % 2:
        assert counts["1"] > 0, "no unit was turned away: the case does not test what it is for"
