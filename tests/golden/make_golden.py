"""Writes the committed regression vectors under tests/golden/ from the CPU oracle.

The reference (AyeTbk/renderer) holds no golden images or fixtures for this path and cannot be run
here (parity unpinned), so these vectors do NOT pin the oracle to the reference.  They
freeze the oracle's present output (coverage, winning triangle, depth bits, RGBA8) on three small
scenes, so that a later change to the raster rules, the shading arithmetic or the shadow cascades — in
the oracle or in the HIP path — shows up as a diff against a committed file rather than only as a
GPU-vs-oracle mismatch.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from renderer_amd.scenes import SceneDesc, c1_sphere  # noqa: E402
from renderer_amd.types import (Camera, Color, Light, Mesh, Submesh, from_rotation_x, from_rotation_y,  # noqa: E402
                                from_scale, from_translation)
from tests.helpers import make_vertices, oracle_server  # noqa: E402


def soup(seed, w, h):
    r = np.random.default_rng(seed)
    s = SceneDesc(f"soup{seed}", w, h, Camera(aspect_ratio=w / h), from_translation((0, 0, -1.0)) @ from_rotation_y(0.1))
    for m in range(3):
        ctr = r.uniform((-2.5, -2.0, 0.3), (2.5, 2.0, 8.0), size=(40, 3))
        pos = (ctr[:, None, :] + r.normal(scale=0.6, size=(40, 3, 3))).reshape(-1, 3)
        nrm = r.normal(size=(120, 3))
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        s.meshes.append(Mesh([Submesh(make_vertices(pos, nrm), np.arange(120, dtype=np.uint32))]))
        s.instances.append((m + 1, from_translation(r.uniform(-0.2, 0.2, 3)), m))
    for k in range(3):
        s.lights.append((50 + k, from_translation(r.uniform((-2, -2, 1), (2, 2, 5))),
                         Light.point(5.0).with_color(Color(*r.uniform(0.2, 1, 3), 2.0))))
    return s


def plate_over_floor(w, h):
    s = SceneDesc("shadow", w, h, Camera(aspect_ratio=w / h), from_translation((0, 3, -4)) @ from_rotation_x(0.5))
    s.meshes.append(Mesh.quad())
    up = from_rotation_x(math.pi / 2)
    s.instances.append((1, from_translation((0, -1, 3)) @ up @ from_scale((9, 9, 1)), 0))
    s.instances.append((2, from_translation((0, 0.2, 3)) @ up @ from_scale((1.6, 1.6, 1)), 0))
    s.lights.append((9, from_rotation_x(1.2), Light.directional().with_color(Color(1.0, 0.9, 0.8, 3.5))))
    return s


def sphere_small():
    s = c1_sphere()
    s.width, s.height = 160, 90
    return s


SCENES = {"soup_seed5_96x64": lambda: soup(5, 96, 64), "plate_over_floor_160x100": lambda: plate_over_floor(160, 100),
          "c1_sphere_160x90": sphere_small}


def render(scene, server):
    scene.populate(server)
    server.render()
    return {"ids": server.read_triangle_ids(), "depth_bits": server.read_depth().view(np.uint32),
            "rgba8": server.read_color_rgba8()}


if __name__ == "__main__":
    for name, make in SCENES.items():
        s = make()
        out = render(s, oracle_server(s.width, s.height, threads=1))
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", name + ".npz"), **out)
        print(name, {k: v.shape for k, v in out.items()}, "covered", int((out["ids"] != 0xFFFFFFFF).sum()))
