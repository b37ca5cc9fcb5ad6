"""Host logic: the product's host math equals the oracle's bit for bit, and both equal f64 numpy."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

from renderer_amd.sort_first import (detile_numpy, owned_tiles, pack_numpy, unpack_numpy)
from renderer_amd.types import (Camera, from_rotation_x, from_rotation_y, from_scale, from_translation, identity,
                                to_cols_array)
from tests.helpers import ROOT, oracle_server


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("probe") / "host_math_probe")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", os.path.join(ROOT, "tests", "host_math_probe.cpp"),
                           "-o", exe])
    return exe


def run_probe(exe, t, cam):
    args = [repr(float(v)) for v in to_cols_array(t)] + [repr(float(np.float32(v))) for v in
                                                         (cam.vfov, cam.aspect_ratio, cam.near, cam.far)]
    out = subprocess.check_output([exe] + args).decode().split("\n")
    mats = [np.array([float.fromhex(x) for x in line.split()], dtype=np.float32) for line in out if line.strip()]
    return mats


CASES = [
    (identity(), Camera()),
    (from_translation((3.0, 2.0, -3.0)) @ from_rotation_y(-0.8), Camera(aspect_ratio=16 / 9)),
    (from_translation((0.3, -7.0, 12.5)) @ from_rotation_x(0.4) @ from_rotation_y(2.1) @ from_scale((1.0, 2.0, 0.5)),
     Camera(vfov=0.7, aspect_ratio=1.25, near=0.3, far=500.0)),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_product_host_math_matches_oracle_bitwise(probe, case):
    t, cam = CASES[case]
    proj, view, pv, inv, look, ortho = run_probe(probe, t, cam)[:6]
    vs = oracle_server(64, 64, threads=1)
    vs.set_camera(t, cam)
    P, V, PV = (np.zeros(16, np.float32) for _ in range(3))
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    vs.api.get_matrices(vs._ctx, fp(P), fp(V), fp(PV))
    assert (P.view(np.uint32) == proj.view(np.uint32)).all()
    assert (V.view(np.uint32) == view.view(np.uint32)).all()
    assert (PV.view(np.uint32) == pv.view(np.uint32)).all()


@pytest.mark.parametrize("case", range(len(CASES)))
def test_host_math_against_float64(probe, case):
    t, cam = CASES[case]
    proj, view, pv, inv, look, ortho = [m.reshape(4, 4).T.astype(np.float64) for m in run_probe(probe, t, cam)[:6]]
    h = 1.0 / math.tan(0.5 * cam.vfov)
    r = cam.far / (cam.far - cam.near)
    want = np.array([[h / cam.aspect_ratio, 0, 0, 0], [0, h, 0, 0], [0, 0, r, -r * cam.near], [0, 0, 1, 0]])
    assert np.allclose(proj, want, rtol=1e-6, atol=1e-7)
    assert np.allclose(view, np.linalg.inv(np.asarray(t, dtype=np.float64)), rtol=1e-5, atol=1e-5)
    assert np.allclose(inv @ pv, np.eye(4), atol=2e-3)
    # look_to_lh(eye = translation, dir = z axis): maps eye to the origin and dir to +Z
    eye = np.asarray(t, dtype=np.float64)[:3, 3]
    d = np.asarray(t, dtype=np.float64)[:3, 2]
    assert np.allclose(look @ np.append(eye, 1.0), [0, 0, 0, 1], atol=1e-4)
    assert np.allclose((look @ np.append(d / np.linalg.norm(d), 0.0))[:3], [0, 0, 1], atol=1e-5)
    # orthographic_lh maps (l,b,n) -> (-1,-1,0) and (r,t,f) -> (1,1,1)
    l, rr, b, tt, n, f = -cam.vfov, cam.aspect_ratio, -cam.near, cam.far, cam.near, cam.far
    assert np.allclose(ortho @ [l, b, n, 1], [-1, -1, 0, 1], atol=1e-5)
    assert np.allclose(ortho @ [rr, tt, f, 1], [1, 1, 1, 1], atol=1e-5)


@pytest.mark.parametrize("case", range(len(CASES)))
def test_shadow_cascades_match_oracle_and_bound_their_slices(probe, case):
    """compute_shadow_cascade_projviews (visual_server.rs:771-863): product host code == oracle bit for
    bit, and each cascade's box holds its slice of the camera frustum (x, y in [-1,1], z in [0,1])."""
    from renderer_amd.types import Light
    t, cam = CASES[case]
    mats = run_probe(probe, t, cam)
    inv, got = mats[3].reshape(4, 4).T.astype(np.float64), mats[6:10]
    vs = oracle_server(64, 64, threads=1)
    vs.set_camera(t, cam)
    tm = np.asarray(t, dtype=np.float32)
    light_t = np.eye(4, dtype=np.float32)
    light_t[:3, 2] = -tm[:3, 1] + np.float32(0.5) * tm[:3, 2]      # z axis of the light = direction the probe used
    vs.set_light(7, light_t, Light.directional())
    want = np.zeros(64, np.float32)
    assert vs.api.get_light_cascades(vs._ctx, 7, want.ctypes.data_as(C.POINTER(C.c_float))) == 0
    for k in range(4):
        assert (want[16 * k:16 * k + 16].view(np.uint32) == got[k].view(np.uint32)).all(), k
    ratios = [(0.0, 0.05), (0.05, 0.1), (0.1, 0.3), (0.3, 1.0)]
    for k, (r0, r1) in enumerate(ratios):
        m = got[k].reshape(4, 4).T.astype(np.float64)
        for sx in (-1, 1):
            for sy in (-1, 1):
                n = inv @ [sx, sy, 0, 1]; f = inv @ [sx, sy, 1, 1]
                n, f = n / n[3], f / f[3]
                for r in (r0, r1):
                    p = m @ (n + (f - n) * r)
                    assert abs(p[3] - 1) < 1e-5
                    assert -1 - 2e-3 <= p[0] <= 1 + 2e-3 and -1 - 2e-3 <= p[1] <= 1 + 2e-3, (k, p)
                    assert -2e-3 <= p[2] <= 1 + 2e-3, (k, p)


def test_f16_clear_colour(probe):
    line = run_probe(probe, identity(), Camera())[10]
    assert [float(v) for v in line] == [float(np.float16(v)) for v in (0.02, 0.04, 0.08)]


def test_tile_partition_is_a_partition():
    for (w, h) in ((3840, 2160), (1920, 1080), (203, 157)):
        tx, ty = (w + 31) // 32, (h + 31) // 32
        for world in (1, 2, 3, 4, 8):
            seen = np.zeros(tx * ty, dtype=int)
            for rank in range(world):
                tiles = owned_tiles(rank, world, tx, ty)
                assert tiles == sorted(tiles)
                seen[tiles] += 1
                vs = oracle_server(w, h, threads=1)
                vs.set_tile_partition(rank, world)
                info = vs.tile_info()
                assert info == {"tile_px": 32, "n_owned": len(tiles), "n_total": tx * ty}
            assert (seen == 1).all()


def test_pack_unpack_roundtrip_and_ordering():
    rng = np.random.default_rng(0)
    depth = rng.uniform(0, 1, size=(7, 9)).astype(np.float32)
    depth[0, 0] = 1.0
    rgba = rng.integers(0, 256, size=(7, 9, 4), dtype=np.uint8)
    p = pack_numpy(depth, rgba)
    assert (p >= 0).all()                       # signed view orders like the unsigned key
    d2, c2 = unpack_numpy(p)
    assert (d2 == depth).all() and (c2 == rgba).all()
    q = pack_numpy(np.minimum(depth * 0.5, 1.0).astype(np.float32), rgba)
    assert (np.minimum(p, q) == q).all()        # nearer depth wins the MIN


def test_detile_numpy():
    w, h, tp = 70, 45, 32
    tx, ty = 3, 2
    frame = np.zeros((h, w, 4), np.uint8)
    tiles = np.zeros((tx * ty, tp, tp, 4), np.uint8)
    for k in range(tx * ty):
        tiles[k] = k + 1
    detile_numpy(frame, tiles, list(range(tx * ty)), tp, tx)
    assert frame[0, 0, 0] == 1 and frame[0, 69, 0] == 3 and frame[44, 0, 0] == 4 and frame[44, 69, 0] == 6


def test_batched_instances_equal_single_calls():
    """rnd_set_mesh_instances is the per-instance call in a loop: same draw order, same
    update-in-place by id - checked on the oracle, which shares the entry point's contract."""
    import numpy as np
    from renderer_amd.types import Camera, Mesh, from_scale, from_translation, identity
    from tests.helpers import oracle_server
    rng = np.random.default_rng(5)
    ids = np.arange(1, 41, dtype=np.uint64)
    ts = [from_translation(tuple(rng.uniform((-2, -1.5, 2), (2, 1.5, 6)))) @ from_scale((0.4, 0.4, 1)) for _ in ids]
    imgs = []
    for batched in (False, True):
        vs = oracle_server(96, 64, threads=2)
        vs.set_camera(identity(), Camera(aspect_ratio=1.5))
        h = vs.asset_server.add(Mesh.quad())
        if batched:
            vs.set_mesh_instances(ids, ts, h)
            vs.set_mesh_instances(ids[:5], ts[5:10], h)          # update in place: order unchanged
        else:
            for i, t in zip(ids, ts):
                vs.set_mesh_instance(int(i), t, h)
            for i, t in zip(ids[:5], ts[5:10]):
                vs.set_mesh_instance(int(i), t, h)
        vs.render()
        imgs.append((vs.read_triangle_ids(), vs.read_depth()))
    assert (imgs[0][0] == imgs[1][0]).all() and (imgs[0][1] == imgs[1][1]).all()
    assert (imgs[0][0] != 0xFFFFFFFF).sum() > 500


def test_reset_scene_puts_the_cascade_camera_back_to_identity():
    """reset_scene replaces RenderScene with its Default (visual_server.rs:579-581, :896-898), so a light set
    after it and before the next set_camera takes its cascades from the identity matrix again."""
    from renderer_amd.types import Light
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    fresh, used = oracle_server(64, 64, threads=1), oracle_server(64, 64, threads=1)
    used.set_camera(CASES[1][0], CASES[1][1])
    light_t = from_rotation_x(1.1) @ from_rotation_y(0.3)
    used.set_light(7, light_t, Light.directional())
    moved = np.zeros(64, np.float32)
    assert used.api.get_light_cascades(used._ctx, 7, fp(moved)) == 0
    used.reset_scene()
    got, want = np.zeros(64, np.float32), np.zeros(64, np.float32)
    for vs, out in ((used, got), (fresh, want)):
        vs.set_light(7, light_t, Light.directional())
        assert vs.api.get_light_cascades(vs._ctx, 7, fp(out)) == 0
    assert (got.view(np.uint32) == want.view(np.uint32)).all() and (got != moved).any()
