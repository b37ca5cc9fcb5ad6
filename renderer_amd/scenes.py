"""Synthetic scenes for the configurations of BASELINE.json.

All data is generated from fixed seeds (0x5EED0001 + config index); nothing is loaded from disk.
C1' stands in for the reference demo scene, whose assets (data/...) are not in the repository
(reference .gitignore:6): one UV sphere, one directional light as in src/bin/main.rs:93-101.
"""
from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

from .types import (VERTEX_DTYPE, Camera, Color, Light, Mesh, Submesh, from_translation, identity,
                    inverse, look_to_lh)

SEED = 0x5EED0001


@dataclass
class SceneDesc:
    name: str
    width: int
    height: int
    camera: Camera
    camera_transform: np.ndarray
    meshes: List[Mesh] = field(default_factory=list)
    # (node id, transform, mesh index)
    instances: List[Tuple[int, np.ndarray, int]] = field(default_factory=list)
    # (node id, transform, Light)
    lights: List[Tuple[int, np.ndarray, Light]] = field(default_factory=list)

    @property
    def n_triangles(self):
        return sum(sum(len(s.indices) // 3 for s in self.meshes[m].submeshes) for _, _, m in self.instances)

    def populate(self, vs, asset_server=None):
        """Issue the reference's per-frame call sequence (engine.rs:100-165) on a VisualServer."""
        a = asset_server if asset_server is not None else vs.asset_server
        handles = [a.add(m) for m in self.meshes]
        vs.set_camera(self.camera_transform, self.camera)
        for lid, t, light in self.lights:
            vs.set_light(lid, t, light)
        for nid, t, m in self.instances:
            vs.set_mesh_instance(nid, t, handles[m], a)
        return handles


# ---- value noise ---------------------------------------------------------------------------

def _hash2(ix, iy, seed):
    with np.errstate(over="ignore"):
        h = (ix.astype(np.uint32) * np.uint32(0x9E3779B1)) ^ (iy.astype(np.uint32) * np.uint32(0x85EBCA77))
        h ^= np.uint32(seed & 0xFFFFFFFF)
        h ^= h >> np.uint32(15)
        h *= np.uint32(0x2C1B3C6D)
        h ^= h >> np.uint32(12)
        h *= np.uint32(0x297A2D39)
        h ^= h >> np.uint32(15)
    return h.astype(np.float64) / 4294967296.0


def value_noise(x, y, seed, octaves=3):
    """Smooth value noise in [-1, 1], a few octaves; x, y float64 arrays."""
    out = np.zeros_like(x, dtype=np.float64)
    amp, freq, norm = 1.0, 1.0, 0.0
    for o in range(octaves):
        xs, ys = x * freq, y * freq
        ix, iy = np.floor(xs), np.floor(ys)
        fx, fy = xs - ix, ys - iy
        sx, sy = fx * fx * (3 - 2 * fx), fy * fy * (3 - 2 * fy)
        ix, iy = ix.astype(np.int64), iy.astype(np.int64)
        s = seed + 101 * o
        v00, v10 = _hash2(ix, iy, s), _hash2(ix + 1, iy, s)
        v01, v11 = _hash2(ix, iy + 1, s), _hash2(ix + 1, iy + 1, s)
        out += amp * ((v00 * (1 - sx) + v10 * sx) * (1 - sy) + (v01 * (1 - sx) + v11 * sx) * sy)
        norm += amp
        amp *= 0.5
        freq *= 2.0
    return out / norm * 2.0 - 1.0


# ---- mesh builders --------------------------------------------------------------------------

def grid_indices(nx, ny):
    """Two CCW-on-screen (front-facing from -Z) triangles per quad; (nx+1) vertices per row."""
    i, j = np.meshgrid(np.arange(nx, dtype=np.uint32), np.arange(ny, dtype=np.uint32), indexing="xy")
    v00 = (j * (nx + 1) + i).reshape(-1)
    v10, v01, v11 = v00 + 1, v00 + (nx + 1), v00 + (nx + 2)
    return np.stack([v00, v10, v11, v00, v11, v01], axis=1).reshape(-1).astype(np.uint32)


def heightfield_mesh(nx, ny, x0, x1, y0, y1, z_base, amp, seed, uv_rect=(0.0, 1.0, 0.0, 1.0)):
    """Height field z = z_base + amp * noise(x, y) facing -Z, analytic-ish (central difference) normals."""
    xs = np.linspace(x0, x1, nx + 1)
    ys = np.linspace(y0, y1, ny + 1)
    X, Y = np.meshgrid(xs, ys, indexing="xy")
    f = lambda a, b: z_base + amp * value_noise(a * 0.9, b * 0.9, seed)
    Z = f(X, Y)
    e = 1e-3
    dzdx = (f(X + e, Y) - f(X - e, Y)) / (2 * e)
    dzdy = (f(X, Y + e) - f(X, Y - e)) / (2 * e)
    n = np.stack([dzdx, dzdy, -np.ones_like(dzdx)], axis=-1)
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    v = np.zeros((ny + 1) * (nx + 1), dtype=VERTEX_DTYPE)
    v["position"] = np.stack([X, Y, Z], axis=-1).reshape(-1, 3).astype(np.float32)
    v["normal"] = n.reshape(-1, 3).astype(np.float32)
    u0, u1, v0, v1 = uv_rect
    U, V = np.meshgrid(np.linspace(u0, u1, nx + 1), np.linspace(v0, v1, ny + 1), indexing="xy")
    v["uv"] = np.stack([U, V], axis=-1).reshape(-1, 2).astype(np.float32)
    return Mesh([Submesh(v, grid_indices(nx, ny))])


def uv_sphere_mesh(n_lon, n_lat, radius=1.0):
    """UV sphere with 2*n_lon*n_lat triangles (pole quads keep their degenerate triangle)."""
    lon = np.linspace(0.0, 2 * np.pi, n_lon + 1)
    lat = np.linspace(0.0, np.pi, n_lat + 1)          # 0 = +Y pole
    LON, LAT = np.meshgrid(lon, lat, indexing="xy")
    nrm = np.stack([np.sin(LAT) * np.cos(LON), np.cos(LAT), np.sin(LAT) * np.sin(LON)], axis=-1)
    v = np.zeros((n_lat + 1) * (n_lon + 1), dtype=VERTEX_DTYPE)
    v["position"] = (nrm * radius).reshape(-1, 3).astype(np.float32)
    v["normal"] = nrm.reshape(-1, 3).astype(np.float32)
    v["uv"] = np.stack([LON / (2 * np.pi), LAT / np.pi], axis=-1).reshape(-1, 2).astype(np.float32)
    # outward-facing = CCW seen from outside.  With x=cos(lon), z=sin(lon), y up, the winding
    # (v00, v10, v11) of grid_indices is CW from outside, so flip it.
    idx = grid_indices(n_lon, n_lat).reshape(-1, 3)[:, ::-1].reshape(-1)
    return Mesh([Submesh(v, np.ascontiguousarray(idx))])


def _point_lights(rng, n, lo, hi, radius, intensity=2.0, first_id=1000):
    out = []
    for k in range(n):
        pos = rng.uniform(lo, hi)
        col = rng.uniform(0.2, 1.0, size=3)
        light = Light.point(radius).with_color(Color(float(col[0]), float(col[1]), float(col[2]), intensity))
        out.append((first_id + k, from_translation(pos), light))
    return out


# ---- configurations -------------------------------------------------------------------------

def c1_sphere(width=1280, height=720):
    """C1': one UV sphere 128x64 (16,384 triangles), one directional light, 1280x720."""
    cam = Camera(aspect_ratio=width / height)
    s = SceneDesc("C1' sphere 16k tri, 1 directional light", width, height, cam, from_translation((0.0, 0.0, -3.0)))
    s.meshes.append(uv_sphere_mesh(128, 64, 1.0))
    s.instances.append((1, identity(), 0))
    t = inverse(look_to_lh((-1.5, 20.0, -6.0), (0.05, -1.0, 0.2)))    # src/bin/main.rs:95-102
    s.lights.append((100, t, Light.directional().with_color(Color(1.0, 0.9, 0.8, 3.5))))
    return s


def _grid_scene(name, width, height, nx, ny, patches_x, patches_y, n_lights, seed, layers=1, layer_dz=0.5):
    cam = Camera(aspect_ratio=width / height)
    s = SceneDesc(name, width, height, cam, identity())
    rng = np.random.default_rng(seed)
    px, py = nx // patches_x, ny // patches_y
    order = []
    for layer in range(layers):
        for j in range(patches_y):
            for i in range(patches_x):
                x0, x1 = -10.0 + 20.0 * i / patches_x, -10.0 + 20.0 * (i + 1) / patches_x
                y0, y1 = -5.0 + 10.0 * j / patches_y, -5.0 + 10.0 * (j + 1) / patches_y
                uv = (i / patches_x, (i + 1) / patches_x, j / patches_y, (j + 1) / patches_y)
                s.meshes.append(heightfield_mesh(px, py, x0, x1, y0, y1, 6.0 + layer_dz * layer, 0.25,
                                                 seed + 7 * layer, uv))
                order.append(len(s.meshes) - 1)
    if layers > 1:
        order = list(rng.permutation(order))
    for k, m in enumerate(order):
        s.instances.append((k + 1, identity(), int(m)))
    s.lights = _point_lights(rng, n_lights, (-10.0, -5.0, 4.75), (10.0, 5.0, 5.25), 6.0)
    return s


def c2_grid_1m(width=1920, height=1080, scale=1.0):
    """C2: 1000x500 quads = 1,000,000 triangles, 8 point lights, 1920x1080."""
    nx, ny = max(2, int(1000 * scale)), max(2, int(500 * scale))
    return _grid_scene(f"C2 grid {2 * nx * ny} tri, 8 point lights", width, height, nx, ny, 1, 1, 8, SEED + 2)


def c3_grid_10m(width=3840, height=2160, scale=1.0):
    """C3: 2500x2000 quads = 10,000,000 triangles in 16 meshes, 32 point lights, 3840x2160."""
    nx, ny = max(4, int(2500 * scale) // 4 * 4), max(4, int(2000 * scale) // 4 * 4)
    return _grid_scene(f"C3 grid {2 * nx * ny} tri, 32 point lights", width, height, nx, ny, 4, 4, 32, SEED + 3)


def c4_layers_50m(width=7680, height=4320, scale=1.0):
    """C4: 10 stacked height fields of 5M triangles (depth complexity ~10), 8 lights, 7680x4320."""
    nx, ny = max(2, int(2500 * scale)), max(2, int(1000 * scale))
    return _grid_scene(f"C4 10 layers x {2 * nx * ny} tri, 8 point lights", width, height, nx, ny, 1, 1, 8,
                       SEED + 4, layers=10)


def c5_instanced_100m(width=3840, height=2160, n_side=10, n_lon=250, n_lat=200):
    """C5: n_side^3 instances of one 2*n_lon*n_lat-triangle sphere on a lattice, 8 lights."""
    cam = Camera(aspect_ratio=width / height)
    s = SceneDesc(f"C5 {n_side ** 3} instances x {2 * n_lon * n_lat} tri", width, height, cam,
                  from_translation((0.0, 0.0, -4.0)))
    s.meshes.append(uv_sphere_mesh(n_lon, n_lat, 0.45))
    rng = np.random.default_rng(SEED + 5)
    k = 1
    for iz in range(n_side):
        for iy in range(n_side):
            for ix in range(n_side):
                pos = np.array([ix - (n_side - 1) / 2, iy - (n_side - 1) / 2, iz * 1.2 + 2.0])
                # random rotation from a random unit quaternion
                q = rng.normal(size=4)
                q /= np.linalg.norm(q)
                w, x, y, z = q
                r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                              [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                              [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
                t = np.eye(4, dtype=np.float32)
                t[:3, :3] = r.astype(np.float32)
                t[:3, 3] = pos.astype(np.float32)
                s.instances.append((k, t, 0))
                k += 1
    half = n_side / 2
    s.lights = _point_lights(rng, 8, (-half, -half, 0.0), (half, half, n_side * 1.2 + 2.0), 8.0)
    return s
