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


def test_rust_sys_crate_declares_the_header():
    """integration/rust/rnd-sys/src/lib.rs is hand-written (no bindgen here): it must name every entry
    point of include/rnd.h, and nothing the header does not have."""
    text = open(os.path.join(ROOT, "integration", "rust", "rnd-sys", "src", "lib.rs")).read()
    declared = set(re.findall(r"pub fn (rnd_[a-z0-9_]+)\s*\(", text))
    header = set(header_functions())
    assert declared == header, (sorted(header - declared), sorted(declared - header))


def test_header_is_plain_c_and_links(tmp_path):
    """A C99 translation unit that includes rnd.h compiles with -pedantic -Werror, links against the
    product library and gets the documented error convention from rnd_create without a GPU (or a
    context with one)."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "consumer.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "rnd.h"
int main(void) {
    rnd_ctx* ctx = NULL;
    rnd_uibox box = {{0.0f, 0.0f}, {1.0f, 1.0f}, {1.0f, 1.0f, 1.0f, 1.0f}};
    rnd_timings t; rnd_frame_stats s; rnd_vertex v; rnd_submesh_desc d;
    int rc = rnd_create(&ctx, 0u, 0u, 0);                 /* bad size: must fail, with a message */
    (void)box; (void)t; (void)s; (void)v; (void)d;
    if (rc != RND_ERR_INVALID || ctx != NULL || strlen(rnd_last_error(NULL)) == 0) return 1;
    if (strcmp(rnd_backend_name(), "hip-gfx950") != 0) return 2;
    rc = rnd_create(&ctx, 64u, 64u, 0);
    if (rc == RND_OK) { rnd_destroy(ctx); printf("context\n"); }
    else if (rc < 0 && strlen(rnd_last_error(NULL)) > 0) printf("no device: %s\n", rnd_last_error(NULL));
    else return 3;
    return 0;
}
''')
    lib_dir = os.path.dirname(_capi.product_library_path())
    exe = tmp_path / "consumer"
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe), "-L", lib_dir, "-lrnd_amd", f"-Wl,-rpath,{lib_dir}"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
