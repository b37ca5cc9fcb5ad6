"""Host-side domain types with the reference's surface.

reference: src/camera.rs (Camera), src/light.rs (Light, LightKind), src/material.rs (Material,
BillboardMode), src/mesh.rs (Mesh, Submesh), src/color.rs (Color), src/arena.rs (Handle),
src/asset_server.rs (AssetServer::add/get), glam::Affine3A (transforms).
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

VERTEX_DTYPE = np.dtype([("position", "<f4", 3), ("normal", "<f4", 3), ("uv", "<f4", 2)])
assert VERTEX_DTYPE.itemsize == 32   # src/renderer/vertex.rs:3-9


@dataclass(frozen=True)
class Color:
    r: float = 1.0
    g: float = 1.0
    b: float = 1.0
    a: float = 1.0

    def to_array(self):
        return [self.r, self.g, self.b, self.a]


Color.WHITE = Color(1.0, 1.0, 1.0, 1.0)
Color.GRUE = Color(0.02, 0.04, 0.08, 1.0)   # src/color.rs:34 (3D target clear colour)


@dataclass
class Camera:
    """src/camera.rs:4-20"""
    vfov: float = 1.3
    aspect_ratio: float = 1.667
    near: float = 0.05
    far: float = 100.0


@dataclass
class Light:
    """src/light.rs:3-59; kind is 'directional' or 'point' (+ radius)."""
    color: Color = Color.WHITE
    kind: str = "point"
    _radius: float = 1.0

    @staticmethod
    def directional():
        return Light(kind="directional", _radius=0.0)

    @staticmethod
    def point(radius):
        return Light(kind="point", _radius=float(radius))

    def with_color(self, color):
        return Light(color=color, kind=self.kind, _radius=self._radius)

    def radius(self):
        return self._radius if self.kind == "point" else None

    def kind_id(self):
        return 0 if self.kind == "directional" else 1


class BillboardMode:
    Off = 0
    On = 1
    FixedSize = 2


@dataclass
class Image:
    """crates/asset_image/src/image.rs:8-113 — RGBA8 (sRGB) pixels, optionally with the mip chain
    `make_mips` builds; `data` is what `Image::data()` hands to `create_color_texture`
    (visual_server.rs:760-768): the levels tightly packed, largest first."""
    width: int
    height: int
    pixels: np.ndarray                      # (height, width, 4) uint8, level 0
    mips: Optional[np.ndarray] = None       # packed chain, set by make_mips
    level_count: int = 1

    def __post_init__(self):
        self.pixels = np.ascontiguousarray(np.asarray(self.pixels, dtype=np.uint8).reshape(self.height, self.width, 4))

    def make_mips(self):
        """image.rs:34-92 with downsample_bilinear :122-186: square power-of-two images only; each level
        is the 2x2 box average of the one above, taken in a gamma-2.0 approximation of linear light
        ((c/255)^2, mean, sqrt, truncated to u8); alpha averaged linearly."""
        if self.width != self.height or self.width & (self.width - 1):
            raise ValueError(f"can't generate mipmaps on images that aren't square and that have non power of two "
                             f"dimensions: dimensions {self.width}x{self.height}")
        levels = [self.pixels]
        size = self.width
        while size > 1:
            src = levels[-1].astype(np.float32) / np.float32(255.0)
            lin = src.copy()
            lin[..., :3] = src[..., :3] * src[..., :3]                      # powi(2)
            quad = [lin[0::2, 0::2], lin[0::2, 1::2], lin[1::2, 0::2], lin[1::2, 1::2]]
            total = np.zeros_like(quad[0])
            for q in quad:                                                   # sum in the reference's order
                total = total + q
            avg = total / np.float32(4.0)
            out = avg.copy()
            out[..., :3] = np.sqrt(avg[..., :3])                             # powf(1/2)
            levels.append((out * np.float32(255.0)).astype(np.uint8))        # `as u8` truncates
            size //= 2
        self.level_count = len(levels)
        self.mips = np.concatenate([lv.reshape(-1) for lv in levels])
        return self

    @property
    def data(self):
        return self.mips if self.mips is not None else self.pixels.reshape(-1)

    @property
    def mip_level_count(self):
        return self.level_count if self.mips is not None else 1


@dataclass
class Material:
    """src/material.rs:3-19"""
    base_color: Color = Color.WHITE
    base_color_image: Optional["Handle"] = None
    billboard_mode: int = BillboardMode.Off
    unlit: bool = False


@dataclass
class Submesh:
    """src/mesh.rs:9-13; vertices as a VERTEX_DTYPE array (32-byte AoS), indices u32."""
    vertices: np.ndarray
    indices: np.ndarray
    material: Optional["Handle"] = None

    def __post_init__(self):
        v = np.asarray(self.vertices)
        if v.dtype != VERTEX_DTYPE:
            v = np.ascontiguousarray(v, dtype=np.float32).reshape(-1, 8).view(VERTEX_DTYPE).reshape(-1)
        self.vertices = np.ascontiguousarray(v)
        self.indices = np.ascontiguousarray(np.asarray(self.indices, dtype=np.uint32).reshape(-1))


@dataclass
class Mesh:
    submeshes: List[Submesh] = field(default_factory=list)

    @staticmethod
    def quad():
        """src/mesh.rs:16-37"""
        v = np.zeros(4, dtype=VERTEX_DTYPE)
        v["position"] = [(-0.5, 0.5, 0.0), (0.5, 0.5, 0.0), (-0.5, -0.5, 0.0), (0.5, -0.5, 0.0)]
        v["normal"] = (0.0, 0.0, -1.0)
        v["uv"] = [(0, 0), (1, 0), (0, 1), (1, 1)]
        return Mesh([Submesh(v, np.array([0, 2, 1, 1, 2, 3], dtype=np.uint32))])


@dataclass(frozen=True)
class Handle:
    """src/arena.rs:54-57 — typed index into the AssetServer."""
    kind: str
    id: int


class AssetServer:
    """The two calls of src/asset_server.rs the hot path needs: add() and get()."""

    def __init__(self):
        self._assets = {}
        self._next = 1

    def add(self, asset):
        h = Handle(type(asset).__name__, self._next)
        self._next += 1
        self._assets[h] = asset
        return h

    def get(self, handle):
        return self._assets[handle]


# ---- transforms: 4x4 float32 arrays in math (row, column) layout; Affine3A analogues ----------

def _m(a):
    return np.asarray(a, dtype=np.float32).reshape(4, 4)


def identity():
    return np.eye(4, dtype=np.float32)


def from_translation(t):
    m = np.eye(4, dtype=np.float32)
    m[:3, 3] = np.asarray(t, dtype=np.float32)
    return m


def from_scale(s):
    m = np.eye(4, dtype=np.float32)
    s = np.broadcast_to(np.asarray(s, dtype=np.float32), (3,))
    m[0, 0], m[1, 1], m[2, 2] = s
    return m


def from_rotation_x(a):
    c, s = np.float32(np.cos(a)), np.float32(np.sin(a))
    m = np.eye(4, dtype=np.float32)
    m[1, 1], m[1, 2], m[2, 1], m[2, 2] = c, -s, s, c
    return m


def from_rotation_y(a):
    c, s = np.float32(np.cos(a)), np.float32(np.sin(a))
    m = np.eye(4, dtype=np.float32)
    m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, s, -s, c
    return m


def from_rotation_z(a):
    c, s = np.float32(np.cos(a)), np.float32(np.sin(a))
    m = np.eye(4, dtype=np.float32)
    m[0, 0], m[0, 1], m[1, 0], m[1, 1] = c, -s, s, c
    return m


def look_to_lh(eye, direction, up=(0.0, 1.0, 0.0)):
    """glam Affine3A::look_to_lh — a VIEW matrix; invert it to place a node (main.rs:97-102)."""
    eye = np.asarray(eye, dtype=np.float64)
    f = np.asarray(direction, dtype=np.float64)
    f = f / np.linalg.norm(f)
    s = np.cross(np.asarray(up, dtype=np.float64), f)
    s = s / np.linalg.norm(s)
    u = np.cross(f, s)
    m = np.eye(4)
    m[0, :3], m[1, :3], m[2, :3] = s, u, f
    m[:3, 3] = [-s @ eye, -u @ eye, -f @ eye]
    return m.astype(np.float32)


def inverse(m):
    return np.linalg.inv(_m(m).astype(np.float64)).astype(np.float32)


def to_cols_array(m):
    """glam Mat4::to_cols_array(): column-major f32[16]"""
    return np.ascontiguousarray(_m(m).T.reshape(16), dtype=np.float32)
