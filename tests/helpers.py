"""Test-side helpers: loading the CPU oracle (test infrastructure) and building small scenes."""
import ctypes as C
import os
import subprocess

import numpy as np

from renderer_amd import _capi
from renderer_amd.types import VERTEX_DTYPE, AssetServer, Mesh, Submesh
from renderer_amd.visual_server import VisualServer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "liboracle.so")

_P = C.c_void_p
_ORACLE_EXTRA = {
    "set_mode": (C.c_int, [_P, C.c_int]),
    "set_threads": (C.c_int, [_P, C.c_int]),
    "get_matrices": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "get_light_cascades": (C.c_int, [_P, C.c_uint64, C.POINTER(C.c_float)]),
    "read_tiles": (C.c_int, [_P, _P]),
}
_oracle = None


def load_oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_LIB):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
        _oracle = _capi.CApi(ORACLE_LIB, "orc_", extra=_ORACLE_EXTRA, allow_missing=_capi.DEVICE_ONLY)
    return _oracle


def oracle_server(width, height, mode="deferred", threads=None):
    api = load_oracle()
    vs = VisualServer(width, height, AssetServer(), capi=api)
    api.set_mode(vs._ctx, 1 if mode == "forward" else 0)
    api.set_threads(vs._ctx, threads if threads else min(8, os.cpu_count() or 1))
    return vs


def make_vertices(positions, normals=None, uvs=None):
    positions = np.asarray(positions, dtype=np.float32).reshape(-1, 3)
    v = np.zeros(len(positions), dtype=VERTEX_DTYPE)
    v["position"] = positions
    v["normal"] = (0.0, 0.0, -1.0) if normals is None else np.asarray(normals, dtype=np.float32).reshape(-1, 3)
    if uvs is not None:
        v["uv"] = np.asarray(uvs, dtype=np.float32).reshape(-1, 2)
    return v


def tri_mesh(positions, indices=None, normals=None, material=None):
    v = make_vertices(positions, normals)
    if indices is None:
        indices = np.arange(len(v), dtype=np.uint32)
    return Mesh([Submesh(v, np.asarray(indices, dtype=np.uint32), material)])


def ndc_camera_args(width, height):
    """A camera for which view-space (x, y) maps to NDC through a known projection."""
    from renderer_amd.types import Camera
    return Camera(vfov=1.3, aspect_ratio=width / height, near=0.05, far=100.0)


def screen_to_view(px, py, z, width, height, cam):
    """View-space point that projects to framebuffer position (px, py) at view depth z."""
    h = 1.0 / np.tan(0.5 * cam.vfov)
    w = h / cam.aspect_ratio
    nx = px / width * 2.0 - 1.0
    ny = 1.0 - py / height * 2.0
    return (nx * z / w, ny * z / h, z)
