"""The C-ABI library loads and exports every entry point include/rnd.h declares (no compute calls)."""
import ctypes as C
import os
import re

import pytest

from renderer_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "rnd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rnd_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_visual_server_surface():
    names = header_functions()
    for need in ("rnd_create", "rnd_destroy", "rnd_set_camera", "rnd_set_light", "rnd_set_mesh_instance",
                 "rnd_set_sprite", "rnd_reset_scene", "rnd_render", "rnd_set_render_size", "rnd_set_msaa",
                 "rnd_set_tone_mapping", "rnd_set_render_size_factor", "rnd_register_mesh", "rnd_register_material"):
        assert need in names


def test_product_library_exports_every_declared_symbol():
    path = _capi.product_library_path()
    assert os.path.exists(path), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = C.CDLL(path)
    missing = [n for n in header_functions() if not hasattr(lib, n)]
    assert not missing, missing


def test_python_binding_covers_the_header():
    declared = {n[len("rnd_"):] for n in header_functions()}
    bound = set(_capi.SIGNATURES)
    assert declared == bound, (declared - bound, bound - declared)


def test_oracle_exports_the_same_surface(oracle_api):
    lib = oracle_api.lib
    for name in _capi.SIGNATURES:
        if name in _capi.DEVICE_ONLY:
            continue
        assert hasattr(lib, "orc_" + name), name


def test_create_fails_loudly_without_a_gpu_or_with_bad_arguments():
    import torch
    api = _capi.load_product()
    assert api.backend_name() == b"hip-gfx950"
    ctx = C.c_void_p()
    assert api.create(C.byref(ctx), 0, 10, 0) == -1          # RND_ERR_INVALID, before any HIP call
    assert b"size" in api.last_error(None)
    if not torch.cuda.is_available():
        rc = api.create(C.byref(ctx), 64, 64, 0)
        assert rc in (-3, -1) and not ctx.value               # RND_ERR_DEVICE: no fallback path
        assert api.last_error(None)


def test_visual_server_raises_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from renderer_amd import RndError, VisualServer
    with pytest.raises(RndError):
        VisualServer(64, 64)
