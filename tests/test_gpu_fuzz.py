"""A fixed-seed slice of the randomized parity sweep (tools/fuzz_parity.py) in the suite the driver runs: 24 random
scenes — triangle soups of every size, clipping, point and directional lights, a quarter at MSAA x4, render size
factors, both tone maps, 2D content on four in ten — through the HIP path and the CPU oracle, half under each source of
the shading kernel's clip positions and a third with the early-depth setup variant forced.  Bar as everywhere: winning
triangle and depth bits equal, colour above 1 LSB on at most 0.1 % of the pixels.  And the HIP path against the
independently written brute-force rasterizer (tests/independent_raster.py), without the oracle in between."""
import numpy as np
import pytest

from renderer_amd import VisualServer
from renderer_amd.types import Camera, Mesh, Submesh, identity
from tests import independent_raster as ir
from tests.helpers import make_vertices, oracle_server
from tests.test_independent_raster import build_case, oracle_projview
from tools.fuzz_parity import run_case

pytestmark = pytest.mark.gpu

SEEDS = list(range(20250, 20274))


@pytest.mark.parametrize("seed", SEEDS)
def test_random_scene_against_oracle(seed, monkeypatch):
    monkeypatch.setenv("RND_STORE_CLIP", str(seed % 2))
    if seed % 3 == 0:
        monkeypatch.setenv("RND_EARLY_DEPTH", "1")
    desc, ids_ok, depth_ok, dmax, frac = run_case(seed)
    assert ids_ok and depth_ok, desc
    assert frac <= 1e-3, (desc, dmax, frac)


def test_hip_path_against_the_independent_rasterizer():
    """Coverage, winning triangle and depth bits of the HIP path against the numpy checker on the 40 first cases of
    tests/test_independent_raster.py.  Only the projection * view matrix is taken from the oracle (host matrix code)."""
    compared = 0
    for seed in range(0, 40):
        r = np.random.default_rng(10_000 + seed)
        w, h = int(r.integers(9, 97)), int(r.integers(9, 71))
        cam = Camera(vfov=1.3, aspect_ratio=w / h, near=0.05, far=100.0)
        pos, tris, model = build_case(seed, w, h, cam)
        cpu = oracle_server(w, h, threads=1)
        cpu.set_camera(identity(), cam)
        pv = oracle_projview(cpu)
        cpu.close()
        clip = ir.clip_positions(pv, np.asarray(model, np.float32), pos)
        if ir.needs_clipper(clip)[tris].any():
            continue
        X, Y, Z = ir.snap(clip, w, h)
        ids_c, depth_c, _ = ir.rasterize(X, Y, Z, tris, w, h)
        vs = VisualServer(w, h, device=0)
        handle = vs.asset_server.add(Mesh([Submesh(make_vertices(pos), tris.reshape(-1))]))
        vs.set_camera(identity(), cam)
        vs.set_mesh_instance(1, model, handle)
        vs.render()
        ids_g, depth_g = vs.read_triangle_ids(), vs.read_depth()
        vs.close()
        assert (ids_g == ids_c).all(), f"seed {seed}: winning triangle differs on {(ids_g != ids_c).sum()} pixels"
        assert (depth_g.view(np.uint32) == depth_c.view(np.uint32)).all(), f"seed {seed}: depth bits differ"
        compared += 1
    assert compared >= 36
