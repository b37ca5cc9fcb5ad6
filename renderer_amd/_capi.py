"""ctypes binding of the C ABI declared in include/rnd.h.

The same declarations bind any library that exports the ABI under a symbol prefix; the
product library is ``renderer_amd/csrc/librnd_amd.so`` with the prefix ``rnd_``.  There is no
CPU fallback: if the HIP library is missing, loading raises.
"""
import ctypes as C
import os

RND_NONE = 0xFFFFFFFF

STATUS = {
    0: "RND_OK",
    -1: "RND_ERR_INVALID",
    -2: "RND_ERR_UNKNOWN_ID",
    -3: "RND_ERR_DEVICE",
    -4: "RND_ERR_OOM",
    -5: "RND_ERR_UNSUPPORTED",
    -6: "RND_ERR_NO_FRAME",
    -7: "RND_ERR_TRUNCATED",
}


class RndError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"{STATUS.get(code, code)}: {text}")
        self.code = code


class Vertex(C.Structure):
    """reference: src/renderer/vertex.rs:3-9"""
    _fields_ = [("position", C.c_float * 3), ("normal", C.c_float * 3), ("uv", C.c_float * 2)]


class SubmeshDesc(C.Structure):
    _fields_ = [
        ("vertices", C.c_void_p),
        ("n_vertices", C.c_uint32),
        ("indices", C.c_void_p),
        ("n_indices", C.c_uint32),
        ("material_id", C.c_uint32),
    ]


class Timings(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "transform_ms", "setup_ms", "scan_ms", "bin_ms", "raster_ms", "shade_ms", "frame_ms",
        "shadow_ms", "clear_ms", "cull_ms", "clip_ms")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class FrameStats(C.Structure):
    _fields_ = [
        ("n_vertices", C.c_uint64), ("n_triangles", C.c_uint64), ("n_setup", C.c_uint64),
        ("n_bin_entries", C.c_uint64), ("n_pixels", C.c_uint64), ("n_fragments", C.c_uint64), ("n_draw_items", C.c_uint32),
        ("n_lights", C.c_uint32), ("n_tiles", C.c_uint32), ("overflow", C.c_uint32), ("early_depth", C.c_uint32), ("n_units_hidden", C.c_uint32),
    ]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class UiBox(C.Structure):
    """reference: UiBoxInstance, src/renderer/pipeline2d/uibox_instance.rs:5-11"""
    _fields_ = [("position", C.c_float * 2), ("size", C.c_float * 2), ("color", C.c_float * 4)]


_F16 = C.POINTER(C.c_float)
_P = C.c_void_p

# name -> (restype, argtypes); every entry is declared in include/rnd.h
SIGNATURES = {
    "create": (C.c_int, [C.POINTER(_P), C.c_uint32, C.c_uint32, C.c_int]),
    "destroy": (None, [_P]),
    "last_error": (C.c_char_p, [_P]),
    "backend_name": (C.c_char_p, []),
    "render_size": (C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "set_render_size": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "render_resolution": (C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "render_size_factor": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "set_render_size_factor": (C.c_int, [_P, C.c_float]),
    "msaa_sample_count": (C.c_int, [_P, C.POINTER(C.c_uint32)]),
    "set_msaa": (C.c_int, [_P, C.c_uint32]),
    "tone_mapping": (C.c_int, [_P, C.POINTER(C.c_uint32)]),
    "set_tone_mapping": (C.c_int, [_P, C.c_uint32]),
    "register_mesh": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.POINTER(SubmeshDesc)]),
    "register_material": (C.c_int, [_P, C.c_uint32, _F16, C.c_uint32, C.c_uint32, C.c_uint32]),
    "register_texture": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _P]),
    "set_camera": (C.c_int, [_P, _F16, C.c_float, C.c_float, C.c_float, C.c_float]),
    "set_light": (C.c_int, [_P, C.c_uint64, _F16, _F16, C.c_uint32, C.c_float]),
    "set_mesh_instance": (C.c_int, [_P, C.c_uint64, _F16, C.c_uint32]),
    "set_mesh_instance_ex": (C.c_int, [_P, C.c_uint64, _F16, C.c_uint32, C.c_uint32, C.c_uint32]),
    "set_mesh_instances": (C.c_int, [_P, C.c_uint32, _P, _P, _P]),
    "set_sprite": (C.c_int, [_P, C.c_uint64, _F16, C.c_uint32, _F16]),
    "reset_scene": (C.c_int, [_P]),
    "set_uiboxes": (C.c_int, [_P, C.c_uint32, _P]),
    "set_font_image": (C.c_int, [_P, C.c_uint32, C.c_uint32, _P]),
    "reset_texts": (C.c_int, [_P]),
    "add_text": (C.c_int, [_P, _P, C.c_uint32, C.c_float, C.c_float, C.c_float]),
    "set_depth_fullscreen_texture": (C.c_int, [_P]),
    "set_shadow_map_fullscreen_texture": (C.c_int, [_P, C.c_uint64]),
    "unset_fullscreen_texture": (C.c_int, [_P]),
    "read_overlay_rgba8": (C.c_int, [_P, _P]),
    "render": (C.c_int, [_P]),
    "finish": (C.c_int, [_P]),
    "render_noalloc": (C.c_int, [_P, _P, C.c_uint64]),
    "read_color_rgba8": (C.c_int, [_P, _P]),
    "read_color_hdr": (C.c_int, [_P, _P]),
    "read_depth": (C.c_int, [_P, _P]),
    "read_coverage": (C.c_int, [_P, _P]),
    "read_triangle_ids": (C.c_int, [_P, _P]),
    "last_timings": (C.c_int, [_P, C.POINTER(Timings)]),
    "mean_timings": (C.c_int, [_P, C.POINTER(Timings), C.POINTER(C.c_uint32), C.c_int]),
    "time_frames": (C.c_int, [_P, C.c_uint32]),
    "timing_stats": (C.c_int, [_P, C.POINTER(Timings), C.POINTER(Timings), C.POINTER(C.c_uint32), C.c_int]),
    "frame_stats_get": (C.c_int, [_P, C.POINTER(FrameStats)]),
    "set_tile_partition": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "tile_info": (C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "tiles_device_ptr": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    "set_tiles_output": (C.c_int, [_P, _P, C.c_uint64]),
    "detile": (C.c_int, [_P, C.c_uint32, C.c_uint32, _P]),
    "detile_gathered": (C.c_int, [_P, C.c_uint32, _P, C.c_uint64]),
    "packed_device_ptr": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    "unpack": (C.c_int, [_P, _P]),
    "set_packed_output": (C.c_int, [_P, _P, C.c_uint64]),
    "min_packed": (C.c_int, [_P, _P, C.c_uint32, C.c_uint64, _P, C.c_uint64]),
    "comm_make_id": (C.c_int, [_P]),
    "comm_init": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32]),
    "comm_destroy": (C.c_int, [_P]),
    "gather_tiles": (C.c_int, [_P]),
    "composite_packed": (C.c_int, [_P]),
    "set_stream": (C.c_int, [_P, _P]),
    "color_device_ptr": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint64)]),
}

# entry points that only make sense with device memory; a host-only library may omit them
DEVICE_ONLY = {"mean_timings", "render_noalloc", "time_frames", "timing_stats", "set_packed_output", "min_packed", "comm_make_id", "comm_init", "comm_destroy", "gather_tiles", "composite_packed", "set_tiles_output", "tiles_device_ptr", "detile", "detile_gathered", "packed_device_ptr", "unpack", "set_stream",
               "color_device_ptr"}


class CApi:
    """A loaded library exporting the rnd.h ABI under `prefix`."""

    def __init__(self, path, prefix="rnd_", extra=None, allow_missing=()):
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} not found: build it first (python -c 'import __graft_entry__ as g; g.build()')")
        self.path = path
        self.prefix = prefix
        self.lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        sigs = dict(SIGNATURES)
        if extra:
            sigs.update(extra)
        for name, (res, args) in sigs.items():
            try:
                fn = getattr(self.lib, prefix + name)
            except AttributeError:
                if name in allow_missing:
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)


_PRODUCT = None


def product_library_path():
    # RND_LIBRARY: another build of the same library (A/B timing of two builds in one GPU call); no other backend exists
    return os.environ.get("RND_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "librnd_amd.so")


def load_product():
    """The HIP library; raises loudly when it has not been built (no fallback)."""
    global _PRODUCT
    if _PRODUCT is None:
        _PRODUCT = CApi(product_library_path(), "rnd_")
    return _PRODUCT
