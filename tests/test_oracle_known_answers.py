"""Known-answer tests that pin the CPU oracle.

The reference holds no tests or golden vectors for this path ("parity unpinned"), so every
expected value here is derived by hand from the WGSL text (render_mesh.wgsl, show_texture.wgsl)
and the WebGPU rasterization rules; the derivation is in each test.
"""
import math

import numpy as np
import pytest

from renderer_amd.types import (AssetServer, Camera, Color, Light, Material, Mesh, Submesh, from_rotation_y,
                                from_translation, identity)
from tests.helpers import make_vertices, oracle_server, screen_to_view, tri_mesh

W, H = 64, 48


def srgb8(x):
    x = min(max(x, 0.0), 1.0)
    y = 12.92 * x if x <= 0.0031308 else 1.055 * x ** (1 / 2.4) - 0.055
    return int(y * 255.0 + 0.5)


def reinhard(c):
    lum = 0.2126 * c[0] + 0.7152 * c[1] + 0.0722 * c[2]
    return [v / (lum + 1.0) for v in c]


def server(mode="deferred", w=W, h=H):
    vs = oracle_server(w, h, mode=mode, threads=1)
    cam = Camera(aspect_ratio=w / h)
    vs.set_camera(identity(), cam)
    return vs, cam


def screen_tri(vs, cam, pts, z=2.0, node=1, w=W, h=H, material=None):
    """Add a triangle whose vertices project exactly onto framebuffer positions pts (px, py)."""
    pos = [screen_to_view(px, py, z if np.isscalar(z) else z[k], w, h, cam) for k, (px, py) in enumerate(pts)]
    mesh = vs.asset_server.add(tri_mesh(pos, material=material))
    vs.set_mesh_instance(node, identity(), mesh)
    return pos


# ---- KA1: projection / depth --------------------------------------------------------------

def test_ka1_depth_of_perspective_lh():
    # depth = r*(z-near)/z with r = far/(far-near); default camera near .05 far 100
    vs, cam = server()
    for z_view, node in ((1.0, 1),):
        screen_tri(vs, cam, [(0, 0), (0, H), (W, 0)], z=z_view, node=node)   # CCW with y up
    vs.render()
    d = vs.read_depth()
    r = 100.0 / 99.95
    assert vs.read_coverage()[5, 5] == 1
    assert abs(float(d[5, 5]) - r * 0.95) < 2e-7
    assert abs(float(d[5, 5]) - 0.9504752) < 1e-6


def test_ka1_near_and_far_planes():
    api_vs, cam = server()
    m = np.zeros(16, dtype=np.float32)
    import ctypes as C
    P, V, PV = (np.zeros(16, np.float32) for _ in range(3))
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    api_vs.api.get_matrices(api_vs._ctx, fp(P), fp(V), fp(PV))
    P = P.reshape(4, 4).T
    for z, want in ((0.05, 0.0), (100.0, 1.0)):
        clip = P @ np.array([0, 0, z, 1], dtype=np.float32)
        assert abs(clip[2] / clip[3] - want) < 1e-6
    # glam perspective_lh layout: w = h/aspect, h = 1/tan(fov/2), m[2][3] = 1, m[3][2] = -r*near
    h = 1.0 / math.tan(0.65)
    assert abs(P[1, 1] - h) < 1e-6 and abs(P[0, 0] - h / cam.aspect_ratio) < 1e-6
    assert P[3, 2] == 1.0 and abs(P[2, 3] + (100 / 99.95) * 0.05) < 1e-7


# ---- KA2: coverage, top-left rule ----------------------------------------------------------

def test_ka2_top_left_rule_on_pixel_centres():
    # Right triangle with vertices on pixel CENTRES (10.5,10.5) (10.5,20.5) (20.5,10.5): the
    # vertical edge x=10.5 is a left edge (in), the horizontal edge y=10.5 is a top edge (in), the
    # hypotenuse x+y=31 is neither (pixels with centre exactly on it are out).
    vs, cam = server()
    screen_tri(vs, cam, [(10.5, 10.5), (10.5, 20.5), (20.5, 10.5)])
    vs.render()
    cov = vs.read_coverage()
    ys, xs = np.nonzero(cov)
    assert xs.min() == 10 and ys.min() == 10          # left column and top row are covered
    for j in range(H):
        for i in range(W):
            inside = i >= 10 and j >= 10 and (i + j) < 30    # centre (i+.5)+(j+.5) < 31
            assert cov[j, i] == int(inside), (i, j)


def test_ka2_bottom_right_edges_excluded():
    # Mirror image: vertices (20.5,20.5) (20.5,10.5) (10.5,20.5); the vertical edge x=20.5 is a
    # right edge and the horizontal edge y=20.5 a bottom edge: both excluded.  The hypotenuse
    # x+y=31 now has the interior on its right, so it is a LEFT edge: centres on it are covered.
    vs, cam = server()
    screen_tri(vs, cam, [(20.5, 20.5), (20.5, 10.5), (10.5, 20.5)])
    vs.render()
    cov = vs.read_coverage()
    for j in range(H):
        for i in range(W):
            inside = i < 20 and j < 20 and (i + j) >= 30    # (i+.5)+(j+.5) >= 31
            assert cov[j, i] == int(inside), (i, j)


def test_ka2_shared_edge_covers_each_pixel_once():
    rng = np.random.default_rng(3)
    for trial in range(20):
        vs, cam = server()
        p = rng.uniform(4, 44, size=(4, 2))
        # quad a-b-c-d split along a-c; order the two triangles CCW with y up (= clockwise in fb)
        a, b, c, d = p
        def ccw_up(t):
            (x0, y0), (x1, y1), (x2, y2) = t
            area_fb = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0)
            return t if area_fb < 0 else [t[0], t[2], t[1]]
        t1, t2 = ccw_up([a, b, c]), ccw_up([a, c, d])
        # only use configurations where b and d are on opposite sides of a-c
        s1 = float((c - a)[0] * (b - a)[1] - (c - a)[1] * (b - a)[0])
        s2 = float((c - a)[0] * (d - a)[1] - (c - a)[1] * (d - a)[0])
        if s1 * s2 >= 0:
            continue
        screen_tri(vs, cam, t1, node=1)
        vs.render()
        c1 = vs.read_coverage().astype(int)
        vs.reset_scene()
        screen_tri(vs, cam, t2, node=2)
        vs.render()
        c2 = vs.read_coverage().astype(int)
        vs.reset_scene()
        screen_tri(vs, cam, t1, node=1)
        screen_tri(vs, cam, t2, node=2)
        vs.render()
        cu = vs.read_coverage().astype(int)
        assert (c1 & c2).sum() == 0
        assert ((c1 | c2) == cu).all()
        assert c1.sum() + c2.sum() == cu.sum()


# ---- KA3: culling ---------------------------------------------------------------------------

def test_ka3_quad_visible_from_minus_z_only():
    for cam_z, yaw, visible in ((-2.0, 0.0, True), (2.0, math.pi, False)):
        vs = oracle_server(W, H, threads=1)
        vs.set_camera(from_translation((0, 0, cam_z)) @ from_rotation_y(yaw), Camera(aspect_ratio=W / H))
        quad = vs.asset_server.add(Mesh.quad())
        vs.set_mesh_instance(1, identity(), quad)
        vs.render()
        assert (vs.read_coverage().sum() > 0) == visible


# ---- KA4: depth test is order independent ----------------------------------------------------

def test_ka4_draw_order_does_not_change_image():
    imgs = []
    for order in ((1, 2), (2, 1)):
        vs, cam = server()
        tris = {1: ([(5, 5), (5, 40), (50, 5)], 2.0), 2: ([(10, 2), (10, 45), (60, 20)], [3.0, 1.5, 1.2])}
        for node in order:
            pts, z = tris[node]
            screen_tri(vs, cam, pts, z=z, node=node)
        vs.render()
        imgs.append((vs.read_color_rgba8(), vs.read_depth()))
    assert (imgs[0][0] == imgs[1][0]).all() and (imgs[0][1] == imgs[1][1]).all()


def test_ka4_equal_depth_tie_goes_to_first_drawn():
    vs, cam = server()
    screen_tri(vs, cam, [(5, 5), (5, 40), (50, 5)], node=7)
    screen_tri(vs, cam, [(5, 5), (5, 40), (50, 5)], node=3)    # identical, drawn second
    vs.render()
    ids = vs.read_triangle_ids()
    assert set(np.unique(ids)) == {0, 0xFFFFFFFF}


# ---- KA5-KA9: shading -------------------------------------------------------------------------

SW, SH = 65, 49     # odd: the centre of pixel (32, 24) lies exactly on the optical axis


def shading_scene(light=None, material=None, mode="deferred", tone=1):
    """Big triangle in the plane z = 0 with normal -Z, camera at (0,0,-2) looking +Z."""
    vs = oracle_server(SW, SH, mode=mode, threads=1)
    vs.set_tone_mapping(tone)
    vs.set_camera(from_translation((0, 0, -2.0)), Camera(aspect_ratio=SW / SH))
    mat = vs.asset_server.add(material) if material is not None else None
    v = make_vertices([(-50, -50, 0), (50, -50, 0), (0, 80, 0)], normals=[(0, 0, -1)] * 3)
    # CCW with y up seen from -Z: (-50,-50) -> (50,-50) -> (0,80)
    mesh = vs.asset_server.add(Mesh([Submesh(v, np.array([0, 1, 2], np.uint32), mat)]))
    vs.set_mesh_instance(1, identity(), mesh)
    if light is not None:
        vs.set_light(9, light[0], light[1])
    vs.render()
    return vs


def centre(vs):
    return vs.read_color_hdr()[SH // 2, SW // 2]


def test_ka5_point_light_blinn_phong_value():
    # fragment ~ origin, n = (0,0,-1), camera (0,0,-2), point light (0,0,-1) r=4 colour (1,1,1,2):
    # d = 1, lin = .75, att = .75^2*(3-1.5) = .84375, diff = 1, spec = 1
    # light term = 2*.84375*(1+1) = 3.375 ; ambient = (.3,.5,.9)*.04
    light = (from_translation((0, 0, -1.0)), Light.point(4.0).with_color(Color(1, 1, 1, 2.0)))
    for mode in ("deferred", "forward"):
        vs = shading_scene(light, mode=mode)
        got = centre(vs)
        want = np.array([0.012, 0.020, 0.036]) + 3.375
        assert np.allclose(got, want, rtol=2e-3), (mode, got, want)     # f16 target: 11-bit mantissa
        img = vs.read_color_rgba8()[SH // 2, SW // 2]
        exp = [srgb8(v) for v in reinhard(list(want))]
        assert all(abs(int(a) - b) <= 1 for a, b in zip(img[:3], exp)) and img[3] == 255


def test_ka5_directional_light_value():
    # directional light pointing +Z (transform = identity: z axis = (0,0,1)), colour (1,.9,.8,3.5):
    # diff = dot(n, -L) = 1; h = normalize(-L + V) = (0,0,-1) at the axis -> spec = 1
    light = (identity(), Light.directional().with_color(Color(1.0, 0.9, 0.8, 3.5)))
    vs = shading_scene(light)
    want = np.array([0.012, 0.020, 0.036]) + np.array([1.0, 0.9, 0.8]) * 3.5 * 2.0
    assert np.allclose(centre(vs), want, rtol=2e-3)


def test_ka6_point_light_cutoff_is_exact_zero():
    light = (from_translation((0, 0, -1.0)), Light.point(0.5).with_color(Color(1, 1, 1, 2.0)))   # d=1 > r
    for mode in ("deferred", "forward"):
        vs = shading_scene(light, mode=mode)
        hdr = vs.read_color_hdr()
        cov = vs.read_coverage().astype(bool)
        amb = np.array([0.012, 0.020, 0.036], dtype=np.float32)
        assert np.allclose(hdr[cov], amb, rtol=1e-3)
        assert len(np.unique(hdr[cov].reshape(-1, 3), axis=0)) == 1     # exactly the ambient term everywhere


def test_ka7_alpha_test_discards_colour_and_depth():
    vs = shading_scene(material=Material(base_color=Color(1, 1, 1, 0.49)))
    assert vs.read_coverage().sum() == 0
    assert (vs.read_depth() == 1.0).all()
    vs = shading_scene(material=Material(base_color=Color(1, 1, 1, 0.5)))
    assert vs.read_coverage().sum() > 0


def test_ka8_unlit_material_is_base_colour_only():
    light = (from_translation((0, 0, -1.0)), Light.point(4.0).with_color(Color(1, 1, 1, 2.0)))
    for mode in ("deferred", "forward"):
        vs = shading_scene(light, material=Material(base_color=Color(0.25, 0.5, 0.75, 1.0), unlit=True), mode=mode)
        assert np.allclose(centre(vs), [0.25, 0.5, 0.75], rtol=1e-3)


def test_ka9_tone_mapping():
    # unlit white: (1,1,1) -> Reinhard 1/(1+1) = .5 -> sRGB 188 ; without tone mapping -> 255
    vs = shading_scene(material=Material(unlit=True))
    assert list(vs.read_color_rgba8()[SH // 2, SW // 2]) == [188, 188, 188, 255]
    vs = shading_scene(material=Material(unlit=True), tone=0)
    assert list(vs.read_color_rgba8()[SH // 2, SW // 2]) == [255, 255, 255, 255]
    # background of an empty scene: Color::GRUE (.02,.04,.08) through f16, tone map and sRGB
    f16 = lambda v: float(np.float16(v))
    grue = [f16(0.02), f16(0.04), f16(0.08)]
    for tone, want in ((0, grue), (1, reinhard(grue))):
        vs = oracle_server(W, H, threads=1)
        vs.set_tone_mapping(tone)
        vs.render()
        assert (vs.read_coverage() == 0).all() and (vs.read_depth() == 1.0).all()
        assert list(vs.read_color_rgba8()[0, 0]) == [srgb8(c) for c in want] + [255]
        assert np.allclose(vs.read_color_hdr()[3, 3], grue, rtol=0, atol=0)


# ---- KA10: near-plane clipping -------------------------------------------------------------------

def test_ka10_near_plane_clip_matches_analytic_polygon():
    # Triangle in the plane y_view = -0.5 (a floor strip) running from behind the camera to z=10.
    # After clipping at z = near, its projection is the polygon computed below in float64.
    w, h = 96, 64
    vs = oracle_server(w, h, threads=1)
    cam = Camera(aspect_ratio=w / h)
    vs.set_camera(identity(), cam)
    tri = np.array([(1.0, -0.5, -3.0), (0.0, -0.5, 10.0), (-1.0, -0.5, -3.0)])   # on screen: right, apex, left = CCW
    mesh = vs.asset_server.add(tri_mesh(tri, normals=[(0, 1, 0)] * 3))
    vs.set_mesh_instance(1, identity(), mesh)
    vs.render()
    cov = vs.read_coverage().astype(bool)
    assert cov.sum() > 0
    # analytic: clip against z >= near in view space, project, point-in-polygon at pixel centres
    near = cam.near
    poly = []
    for k in range(3):
        a, b = tri[k], tri[(k + 1) % 3]
        if a[2] >= near:
            poly.append(a)
        if (a[2] >= near) != (b[2] >= near):
            t = (near - a[2]) / (b[2] - a[2])
            poly.append(a + t * (b - a))
    hh = 1.0 / math.tan(0.5 * cam.vfov)
    ww = hh / cam.aspect_ratio
    scr = np.array([((p[0] * ww / p[2] + 1) * 0.5 * w, (1 - p[1] * hh / p[2]) * 0.5 * h) for p in poly])
    ys, xs = np.mgrid[0:h, 0:w]
    cx, cy = xs + 0.5, ys + 0.5
    inside = np.ones((h, w), bool)
    margin = np.full((h, w), np.inf)
    n = len(scr)
    sign = None
    for k in range(n):
        (x0, y0), (x1, y1) = scr[k], scr[(k + 1) % n]
        e = (x1 - x0) * (cy - y0) - (y1 - y0) * (cx - x0)
        if sign is None:
            sign = np.sign(e[(cy > scr[:, 1].min() + 1) & cov].mean())
        inside &= (e * sign) > 0
        margin = np.minimum(margin, np.abs(e) / math.hypot(x1 - x0, y1 - y0))
    # pixels whose centre is more than 1/64 px from every polygon edge must agree exactly
    safe = margin > 1 / 64
    assert (cov[safe] == inside[safe]).all()
    assert np.isfinite(vs.read_depth()).all()
    assert (vs.read_depth()[cov] < 1.0).all() and (vs.read_depth()[cov] >= 0.0).all()


# ---- KA11: the reference's forward multi-pass == the deferred formulation ---------------------------

def test_ka11_forward_equals_deferred_on_random_scene():
    rng = np.random.default_rng(11)
    w, h = 160, 120
    outs = {}
    for mode in ("deferred", "forward"):
        vs = oracle_server(w, h, mode=mode, threads=1)
        cam = Camera(aspect_ratio=w / h)
        vs.set_camera(from_translation((0, 0, -1.0)), cam)
        r = np.random.default_rng(5)
        for m in range(6):
            n = 40
            ctr = r.uniform((-2, -1.5, 2.0), (2, 1.5, 6.0), size=(n, 3))
            pos = (ctr[:, None, :] + r.normal(scale=0.5, size=(n, 3, 3))).reshape(-1, 3)
            nrm = r.normal(size=(n * 3, 3))
            nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
            mat = vs.asset_server.add(Material(base_color=Color(*r.uniform(0.2, 1.0, 3), 1.0), unlit=(m == 5)))
            mesh = vs.asset_server.add(Mesh([Submesh(make_vertices(pos, nrm), np.arange(n * 3, dtype=np.uint32), mat)]))
            vs.set_mesh_instance(m + 1, from_translation(r.uniform(-0.2, 0.2, 3)), mesh)
        for k in range(5):
            p = r.uniform((-2, -2, 1), (2, 2, 5))
            vs.set_light(50 + k, from_translation(p), Light.point(5.0).with_color(Color(*r.uniform(0.2, 1, 3), 2.0)))
        vs.set_light(60, from_rotation_y(0.4), Light.directional().with_color(Color(1.0, 0.9, 0.8, 1.5)))
        vs.render()
        outs[mode] = (vs.read_color_rgba8().astype(int), vs.read_depth(), vs.read_coverage(), vs.read_color_hdr())
    d, f = outs["deferred"], outs["forward"]
    assert (d[2] == f[2]).all() and (d[1] == f[1]).all()
    assert d[2].sum() > 1000
    diff = np.abs(d[0] - f[0]).max(axis=-1)
    assert (diff <= 1).mean() >= 0.999, (diff > 1).mean()
    # HDR: at most a few f16 ulps (one rounding per light pass in the forward target)
    rel = np.abs(d[3] - f[3]) / np.maximum(np.abs(f[3]), 1e-3)
    assert np.percentile(rel, 99.9) < 6 * 2 ** -11


def test_ka12_directional_shadow_cascades():
    """R9 / SURVEY.md 8f N1: a plate above a floor under one directional light darkens the floor
    below it (compute_light_occlusion, render_mesh.wgsl:184-252); with casts_shadows = false on the
    plate (RenderMeshInstance, visual_server.rs:961; pipeline3d.rs:303) the floor is evenly lit."""
    import math
    from renderer_amd.scenes import SceneDesc
    from renderer_amd.types import Camera, Color, Light, Mesh, from_rotation_x, from_scale, from_translation
    w, h = 320, 200
    s = SceneDesc("shadow", w, h, Camera(aspect_ratio=w / h), from_translation((0, 3, -4)) @ from_rotation_x(0.5))
    s.meshes.append(Mesh.quad())
    up = from_rotation_x(math.pi / 2)
    s.instances.append((1, from_translation((0, -1, 3)) @ up @ from_scale((9, 9, 1)), 0))
    s.instances.append((2, from_translation((0, 0.2, 3)) @ up @ from_scale((1.6, 1.6, 1)), 0))
    s.lights.append((9, from_rotation_x(1.2), Light.directional().with_color(Color(1.0, 0.9, 0.8, 3.5))))
    cpu = oracle_server(w, h, threads=4)
    handles = s.populate(cpu)
    cpu.render()
    ids, hdr = cpu.read_triangle_ids(), cpu.read_color_hdr()
    floor = hdr[ids < 2].sum(axis=-1)
    lit = np.median(floor)
    dark = floor < 0.3 * lit
    assert 200 < dark.sum() < 2000
    # fully shadowed floor shows the ambient term only: base * (0.3, 0.5, 0.9) * 0.04, f16-rounded
    amb = np.float16(np.float32([0.3, 0.5, 0.9]) * np.float32(0.04)).astype(np.float32)
    assert np.isclose(floor.min(), amb.sum(), rtol=2e-3)
    cpu.set_mesh_instance(2, s.instances[1][1], handles[0], casts_shadows=False)
    cpu.render()
    ids, hdr = cpu.read_triangle_ids(), cpu.read_color_hdr()
    floor = hdr[ids < 2].sum(axis=-1)
    assert (floor < 0.3 * np.median(floor)).sum() == 0


# ---- KA13/KA14: textured materials ---------------------------------------

def textured_quad_scene(vs, cam, size, image, w, h):
    """A quad that covers the whole w x h frame with uv (0,0) at the top-left pixel corner."""
    from renderer_amd.types import Image  # noqa: F401
    tex = vs.asset_server.add(image)
    mat = vs.asset_server.add(Material(base_color=Color(1, 1, 1, 1), base_color_image=tex, unlit=True))
    corners = [(0, 0), (w, 0), (0, h), (w, h)]
    pos = [screen_to_view(px, py, 2.0, w, h, cam) for px, py in corners]
    uvs = [(0, 0), (1, 0), (0, 1), (1, 1)]
    v = make_vertices(pos, uvs=uvs)
    mesh = vs.asset_server.add(Mesh([Submesh(v, np.array([0, 2, 1, 1, 2, 3], dtype=np.uint32), mat)]))
    vs.set_mesh_instance(1, identity(), mesh)


def test_ka13_texture_one_texel_per_pixel_round_trips():
    """Rgba8UnormSrgb texel -> linear -> unlit, no tone map -> sRGB surface: with one texel per pixel
    (lambda = 0, taps on texel centres) the frame equals the texture bytes."""
    from renderer_amd.types import Image
    n = 16
    r = np.random.default_rng(3)
    px = r.integers(0, 256, (n, n, 4)).astype(np.uint8)
    px[..., 3] = 255
    cam = Camera(aspect_ratio=1.0)
    vs = oracle_server(n, n, threads=1)
    vs.set_tone_mapping(0)
    vs.set_camera(identity(), cam)
    textured_quad_scene(vs, cam, n, Image(n, n, px), n, n)
    vs.render()
    out = vs.read_color_rgba8()
    assert (vs.read_coverage() == 1).all()
    assert np.abs(out.astype(int) - px.astype(int)).max() <= 1
    assert (out == px).mean() > 0.98


def test_ka14_texture_minified_eightfold_reads_mip_3():
    """A 128^2 texture on a 16^2 frame: rho = 8 texels per pixel, lambda = 3, and pixel centres fall on
    the texel centres of level 3 — the frame equals the chain's fourth level (image.rs:34-186)."""
    from renderer_amd.types import Image
    n, frame = 128, 16
    r = np.random.default_rng(4)
    px = r.integers(0, 256, (n, n, 4)).astype(np.uint8)
    px[..., 3] = 255
    img = Image(n, n, px).make_mips()
    assert img.mip_level_count == 8
    off = sum((n >> l) ** 2 * 4 for l in range(3))
    level3 = img.data[off:off + frame * frame * 4].reshape(frame, frame, 4)
    cam = Camera(aspect_ratio=1.0)
    vs = oracle_server(frame, frame, threads=1)
    vs.set_tone_mapping(0)
    vs.set_camera(identity(), cam)
    textured_quad_scene(vs, cam, n, img, frame, frame)
    vs.render()
    out = vs.read_color_rgba8()
    assert np.abs(out.astype(int) - level3.astype(int)).max() <= 1


def test_ka15_texture_alpha_test_per_fragment():
    """base.a = material.a * texel.a < 0.5 discards the fragment (render_mesh.wgsl:89-92): with one texel
    per pixel the coverage mask equals the texture's alpha >= 128 mask, and the pixels behind show
    through (depth 1.0, clear colour)."""
    from renderer_amd.types import Image
    n = 16
    r = np.random.default_rng(8)
    px = r.integers(0, 256, (n, n, 4)).astype(np.uint8)
    px[..., 3] = np.where(r.random((n, n)) < 0.5, 255, 0)
    cam = Camera(aspect_ratio=1.0)
    vs = oracle_server(n, n, threads=1)
    vs.set_tone_mapping(0)
    vs.set_camera(identity(), cam)
    textured_quad_scene(vs, cam, n, Image(n, n, px), n, n)
    vs.render()
    cov = vs.read_coverage()
    assert (cov == (px[..., 3] == 255)).all()
    assert (vs.read_depth()[cov == 0] == 1.0).all()
    out = vs.read_color_rgba8()
    assert np.abs(out[cov == 1].astype(int) - px[cov == 1].astype(int)).max() <= 1
    # material alpha scales it: 0.4 * 1.0 < 0.5 -> nothing is drawn at all
    vs2 = oracle_server(n, n, threads=1)
    vs2.set_camera(identity(), cam)
    tex = vs2.asset_server.add(Image(n, n, px))
    mat = vs2.asset_server.add(Material(base_color=Color(1, 1, 1, 0.4), base_color_image=tex, unlit=True))
    pos = [screen_to_view(x, y, 2.0, n, n, cam) for x, y in [(0, 0), (n, 0), (0, n), (n, n)]]
    mesh = vs2.asset_server.add(Mesh([Submesh(make_vertices(pos, uvs=[(0, 0), (1, 0), (0, 1), (1, 1)]),
                                              np.array([0, 2, 1, 1, 2, 3], dtype=np.uint32), mat)]))
    vs2.set_mesh_instance(1, identity(), mesh)
    vs2.render()
    assert vs2.read_coverage().sum() == 0


# ---- KA16: render_size_factor ----------------------------------------------

@pytest.mark.parametrize("factor", [0.5, 0.25, 2.0])
def test_ka16_render_size_factor(factor):
    """The 3D target is render_size * factor (visual_server.rs:621-623) and is shown on the surface by
    show_texture.wgsl: nearest at factors <= 1 (every surface pixel copies the target pixel under its
    centre), linear above 1 (at factor 2 the centre falls on a corner shared by four target pixels:
    their mean, taken in HDR before the tone map)."""
    w, h = 64, 48
    vs, cam = server(w=w, h=h)
    vs.set_light(5, from_translation((0.3, 0.2, 0.5)), Light.point(6.0).with_color(Color(1.0, 0.8, 0.6, 2.0)))
    screen_tri(vs, cam, [(4, 6), (10, 44), (60, 20)], z=2.0, w=w, h=h)
    vs.set_render_size_factor(factor)
    assert vs.render_size() == (w, h) and vs.render_resolution() == (int(w * factor), int(h * factor))
    vs.render()
    out, hdr = vs.read_color_rgba8(), vs.read_color_hdr()
    assert out.shape == (h, w, 4) and hdr.shape == (int(h * factor), int(w * factor), 3)

    def present(c):
        c = c.astype(np.float64)
        c = c / ((0.2126 * c[..., 0] + 0.7152 * c[..., 1] + 0.0722 * c[..., 2]) + 1.0)[..., None]
        y = np.where(c <= 0.0031308, 12.92 * c, 1.055 * np.clip(c, 1e-12, None) ** (1 / 2.4) - 0.055)
        return np.floor(np.clip(y, 0, 1) * 255.0 + 0.5).astype(int)

    if factor <= 1:
        k = int(round(1 / factor))
        want = present(np.repeat(np.repeat(hdr, k, axis=0), k, axis=1))
    else:
        k = int(factor)
        want = present(hdr.reshape(h, k, w, k, 3).astype(np.float64).mean(axis=(1, 3)))
    assert np.abs(out[..., :3].astype(int) - want).max() <= 1
    assert (out[..., 3] == 255).all()
    assert (vs.read_coverage().sum() > 50 * factor * factor)


# ---- KA17: MSAA x4 ---------------------------------------------------------

def test_ka17_msaa4_edge_coverage_fractions():
    """Sample count 4 (visual_server.rs:174-181): coverage per sample at the standard pattern
    (6,2) (14,6) (2,10) (10,14) /16, one shading per pixel, resolve = mean of the samples.  A unlit
    white half-plane whose vertical edge cuts pixel columns at x = k + f covers the samples with
    offset.x > f: the resolved value steps through 4/4, 3/4, 2/4, 1/4, 0/4 of white over the clear
    colour."""
    w, h = 32, 8
    vs = oracle_server(w, h, threads=1)
    cam = Camera(aspect_ratio=w / h)
    vs.set_camera(identity(), cam)
    vs.set_tone_mapping(0)
    vs.set_msaa(4)
    assert vs.msaa_sample_count() == 4
    mat = vs.asset_server.add(Material(base_color=Color(1, 1, 1, 1), unlit=True))
    clear = np.float32([np.float16(0.02), np.float16(0.04), np.float16(0.08)])
    for edge, n_cov in ((10.0, 4), (10.25, 3), (10.5, 2), (10.75, 1), (10.9, 0)):
        # the half-plane x >= edge, as one big triangle pair spanning the rows
        pts = [(edge, -4), (edge, h + 4), (w + 8, -4), (w + 8, h + 4)]
        pos = [screen_to_view(x, y, 2.0, w, h, cam) for x, y in pts]
        mesh = vs.asset_server.add(Mesh([Submesh(make_vertices(pos), np.array([0, 1, 2, 2, 1, 3], dtype=np.uint32), mat)]))
        vs.set_mesh_instance(1, identity(), mesh)
        vs.render()
        hdr = vs.read_color_hdr()
        want = np.float16((n_cov * np.float32(1.0) + (4 - n_cov) * clear) * np.float32(0.25)).astype(np.float32)
        assert np.allclose(hdr[3, 10], want, rtol=0, atol=1e-6), (edge, hdr[3, 10], want)
        assert np.allclose(hdr[3, 11], 1.0) and np.allclose(hdr[3, 9], clear)


# ---- KA18-KA21: the 2D overlay (pipeline2d.rs, uibox.wgsl, text.wgsl, fullscreen_texture.wgsl) ----

def srgb_decode(v):
    x = v / 255.0
    return x / 12.92 if x <= 0.04045 else ((x + 0.055) / 1.055) ** 2.4


def test_ka18_uibox_coverage_and_blend():
    # A box at (3, 5) of size (4, 2) covers pixel centres with 3 <= x < 7, 5 <= y < 7 (y down, origin at
    # the top-left: translation = viewport.y - pos.y - size.y in the y-up space of uibox.wgsl:38).
    # Opaque: colour = pow(c, 2.2) (uibox.wgsl:27) stored to Rgba8UnormSrgb, alpha 255; the composite
    # over the 3D image (backend.rs:484-490) with alpha 1 replaces it.
    vs = oracle_server(W, H, threads=1)
    vs.set_uiboxes([((3.0, 5.0), (4.0, 2.0), (0.5, 0.25, 1.0, 1.0))])
    vs.render()
    ov = vs.read_overlay_rgba8()
    mask = ov[..., 3] != 0
    want = np.zeros((H, W), bool)
    want[5:7, 3:7] = True
    assert (mask == want).all()
    px = [srgb8(0.5 ** 2.2), srgb8(0.25 ** 2.2), 255, 255]
    assert list(ov[5, 3]) == px and list(ov[6, 6]) == px
    frame = vs.read_color_rgba8()
    assert list(frame[5, 3]) == px
    f16 = lambda v: float(np.float16(v))
    background = [srgb8(c) for c in reinhard([f16(0.02), f16(0.04), f16(0.08)])] + [255]
    assert list(frame[4, 3]) == background and list(frame[5, 7]) == background
    # half-open on a fractional edge: a box from x = 3.5 covers the centre of pixel 3 (3.5 >= 3.5, the
    # left edge is inclusive) and its right edge at 5.5 excludes the centre of pixel 5
    vs.set_uiboxes([((3.5, 0.0), (2.0, 1.0), (1.0, 1.0, 1.0, 1.0))])
    vs.render()
    assert list(np.nonzero(vs.read_overlay_rgba8()[0, :, 3])[0]) == [3, 4]
    # a negative size on one axis turns the strip clockwise: culled (pipeline2d.rs:508-509)
    vs.set_uiboxes([((10.0, 10.0), (-4.0, 4.0), (1.0, 1.0, 1.0, 1.0))])
    vs.render()
    assert not vs.read_overlay_rgba8().any()


def test_ka18_translucent_boxes_blend_in_order_with_8bit_stores():
    # Two boxes over each other, alpha .5 each, ALPHA_BLENDING into the transparent 2D target, which
    # stores 8 bits after each draw; then the 2D target over the 3D image with its own alpha again.
    vs = oracle_server(W, H, threads=1)
    vs.set_tone_mapping(0)
    vs.set_uiboxes([((0.0, 0.0), (8.0, 8.0), (1.0, 0.0, 0.0, 0.5)), ((4.0, 4.0), (8.0, 8.0), (0.0, 1.0, 0.0, 0.5))])
    vs.render()
    ov = vs.read_overlay_rgba8()
    # first box alone: rgb = 1*.5 + 0*.5 = .5 -> 188; alpha = .5 -> round(127.5 + .5) = 128
    assert list(ov[1, 1]) == [188, 0, 0, 128]
    # overlap: red' = decode(188)*.5, green = .5, alpha = .5 + (128/255)*.5
    r = srgb8(srgb_decode(188) * 0.5)
    a = int((0.5 + (128 / 255.0) * 0.5) * 255.0 + 0.5)
    assert list(ov[5, 5]) == [r, 188, 0, a]
    # composite: out = decode(2D) * a2d + decode(3D) * (1 - a2d), background GRUE without tone mapping
    f16 = lambda v: float(np.float16(v))
    bg = [srgb8(f16(0.02)), srgb8(f16(0.04)), srgb8(f16(0.08))]
    a2 = 128 / 255.0
    want = [srgb8(srgb_decode(s) * a2 + srgb_decode(b) * (1.0 - a2)) for s, b in zip([188, 0, 0], bg)]
    assert list(vs.read_color_rgba8()[1, 1]) == want + [255]


def test_ka19_text_glyph_quads_with_the_default_font():
    # add_text (visual_server.rs:556-577): glyph i at x + i*fs*1.1667*.5, size (fs*1.1667*.5, fs).
    # The default font texture is one (255,255,0,255) texel (:82): b = 0, so fill_dist = outline_dist = 0
    # (text.wgsl:95-96), colour = mix(0, 1, 1) = 1, alpha = 1: opaque white cells.
    vs = oracle_server(W, H, threads=1)
    fs = 12.0
    vs.add_text(0, "ab", (2.0, 3.0), fs)
    vs.render()
    ov = vs.read_overlay_rgba8()
    adv = np.float32(fs) * np.float32(1.1667) * np.float32(0.5)          # 7.0002
    x_end = float(np.float32(2.0) + np.float32(1.0) * np.float32(fs) * np.float32(1.1667) * np.float32(0.5) + adv)
    cols = [i for i in range(W) if 2.0 <= i + 0.5 < x_end]
    want = np.zeros((H, W), bool)
    want[3:15, cols[0]:cols[-1] + 1] = True
    assert ((ov[..., 3] == 255) == want).all()
    assert (ov[want] == 255).all()
    vs.reset_texts()
    vs.render()
    assert not vs.read_overlay_rgba8().any()


def test_ka19_sdf_font_sampling():
    # A 256x1 atlas (128 cells of 2 texels): only texel 130, the first of glyph 65's cell, has b = 255.
    # The 24 px glyph quad is 14 px wide (24*1.1667*.5 = 14.0004, snapped to 14): u = (x + .5 - 10)/14,
    # atlas coordinate 130 + 2u - .5 (Repeat + linear sampler, text.wgsl:58-63).  The centre of pixel 13
    # (u = .25) sits on texel 130: b = 1, fill_dist = outline_dist = 1 (smoothing .15 at font size 24),
    # alpha 0 - nothing drawn.  From pixel 20 on (u >= .75) only texels with b = 0 are read: opaque white.
    # Pixel 10 blends texels 129 and 130 with weight .571: below the outline's lower edge .65 (alpha 1) and
    # above the fill's upper edge .52 (fill_alpha 0): the black outline.
    from renderer_amd.types import Image
    atlas = np.zeros((1, 256, 4), np.uint8)
    atlas[0, 130, 2] = 255
    vs = oracle_server(W, H, threads=1)
    img = vs.asset_server.add(Image(256, 1, atlas))
    vs.set_font_image(img)
    vs.add_text(0, "A", (10.0, 4.0), 24.0)
    vs.render()
    ov = vs.read_overlay_rgba8()
    assert (ov[4:28, 13, 3] == 0).all() and (ov[4:28, 13] == 0).all()
    assert (ov[4:28, 20:24] == 255).all() and (ov[4:28, 10] == [0, 0, 0, 255]).all()
    assert not ov[:4].any() and not ov[28:].any() and not ov[:, :10].any() and not ov[:, 24:].any()
    row = ov[10, :, 3].astype(int)
    assert (np.diff(row[10:14]) <= 0).all() and (np.diff(row[13:21]) >= 0).all()
    # pixel 12: u = 2.5/14, b = frac(130 + 5/14 - .5) = .857 (texels 129|130): outline t = (.857-.65)/.3
    t = (6.0 / 7.0 - 0.65) / 0.3
    assert abs(row[12] - (1.0 - t * t * (3.0 - 2.0 * t)) * 255.0) <= 1.0


def test_ka20_depth_fullscreen_texture():
    # fullscreen_texture.wgsl: grey = (2*.5)/(50.5 - d*49.5), alpha 1, under the boxes and texts.
    vs, cam = server()
    screen_tri(vs, cam, [(8, 8), (8, 40), (40, 8)], z=2.0)
    vs.set_depth_fullscreen_texture()
    vs.render()
    d = vs.read_depth()
    ov = vs.read_overlay_rgba8()
    assert (ov[..., 3] == 255).all()
    for (j, i) in ((0, 0), (12, 12), (30, 10)):
        dd = np.float32(1.0) / (np.float32(50.5) - np.float32(d[j, i]) * np.float32(49.5))
        assert list(ov[j, i]) == [srgb8(float(dd))] * 3 + [255]
        assert list(vs.read_color_rgba8()[j, i]) == [srgb8(float(dd))] * 3 + [255]
    assert d[0, 0] == 1.0 and list(ov[0, 0][:3]) == [255, 255, 255]
    vs.unset_fullscreen_texture()
    vs.render()
    assert not vs.read_overlay_rgba8().any()
    # reset_scene drops it with the rest of RenderScene (visual_server.rs:579-581, :903)
    vs.set_depth_fullscreen_texture()
    vs.reset_scene()
    vs.render()
    assert not vs.read_overlay_rgba8().any()


def test_ka21_shadow_map_fullscreen_texture():
    # layer 0 of the light's cascade maps, stretched over the surface with the nearest sampler; an
    # unknown light changes nothing (visual_server.rs:329-332)
    vs, cam = server()
    screen_tri(vs, cam, [(8, 8), (8, 40), (40, 8)], z=2.0)
    vs.set_light(7, from_rotation_y(0.3), Light.directional().with_color(Color(1, 1, 1, 2)))
    vs.set_shadow_map_fullscreen_texture(99)
    vs.render()
    assert not vs.read_overlay_rgba8().any()
    vs.set_shadow_map_fullscreen_texture(7)
    vs.render()
    ov = vs.read_overlay_rgba8()
    assert (ov[..., 3] == 255).all()
    # cleared map texels show (2*.5)/(50.5 - 49.5) = 1 -> white; the triangle's texels are darker
    assert (ov[..., 0] == 255).any() and (ov[..., 0] < 255).any()
    assert (ov[..., 0] == ov[..., 1]).all() and (ov[..., 1] == ov[..., 2]).all()


# ---- KA22: billboard modes through a material (render_mesh.wgsl:58-79) ------------------------

def blinn_phong_point(frag, n, cam, light_pos, radius, color, intensity):
    """render_mesh.wgsl:134-181, :256-281 for one point light, in f64 (white base colour)."""
    frag, n, cam, light_pos = (np.asarray(v, dtype=np.float64) for v in (frag, n, cam, light_pos))
    n = n / np.linalg.norm(n)
    d = np.linalg.norm(frag - light_pos)
    if d > radius:
        return np.zeros(3)
    t = min(max((radius - d) / radius, 0.0), 1.0)
    att = t * t * (3.0 - 2.0 * t)
    ldir = (frag - light_pos) / d
    view = (cam - frag) / np.linalg.norm(cam - frag)
    diff = max(np.dot(n, -ldir), 0.0)
    h = (-ldir + view) / np.linalg.norm(-ldir + view)
    spec = max(np.dot(n, h), 0.0) ** 8
    return np.asarray(color, dtype=np.float64) * intensity * att * (diff + spec)


def billboard_scene(mode, unlit, model, w, h, light_pos=None):
    vs = oracle_server(w, h, threads=1)
    vs.set_camera(identity(), Camera(aspect_ratio=w / h))
    mat = vs.asset_server.add(Material(billboard_mode=mode, unlit=unlit))
    quad = Mesh.quad()
    quad.submeshes[0].material = mat
    vs.set_mesh_instance(1, model, vs.asset_server.add(quad))
    if light_pos is not None:
        vs.set_light(9, from_translation(light_pos), Light.point(4.0).with_color(Color(1, 1, 1, 2.0)))
    vs.render()
    return vs


def test_ka22_billboard_fixed_size_covers_a_constant_ndc_square():
    # mode 2: clip = (ndc(model origin).xy + pos.xy * 0.4, ndc.z, 1): the unit quad always covers the NDC
    # square of half-size 0.2 around the projected origin, whatever the model's rotation, scale or distance.
    # 100x50 frame, origin on the axis: x_fb in [40, 60), y_fb in [20, 30) -> exactly 20 x 10 pixels.
    from renderer_amd.types import from_scale
    w, h = 100, 50
    r = 100.0 / 99.95
    for z, model_rs in ((4.0, from_rotation_y(math.pi) @ from_scale(3.0)), (9.0, from_rotation_y(0.7) @ from_scale((0.1, 5.0, 2.0)))):
        vs = billboard_scene(2, True, from_translation((0, 0, z)) @ model_rs, w, h)
        cov = vs.read_coverage()
        want = np.zeros((h, w), np.uint8)
        want[20:30, 40:60] = 1
        assert (cov == want).all()
        d = vs.read_depth()[cov == 1]
        assert np.allclose(d, r * (z - 0.05) / z, atol=2e-7) and len(np.unique(d)) == 1      # depth of the origin, flat
        assert np.allclose(vs.read_color_hdr()[25, 50], [1, 1, 1], rtol=1e-3)                 # unlit white


def test_ka22_billboard_fixed_size_lit_keeps_the_vertex_normal_and_the_model_position():
    # lit mode 2: normal = vertex.normal (0,0,-1) although the model matrix turns it away from the camera
    # (render_mesh.wgsl:78); frag_pos stays model * pos (:52).  Pixel (50, 25): centre (50.5, 25.5) = NDC
    # (0.01, -0.02) = origin + pos.xy * 0.4 -> pos = (0.025, -0.05, 0); w = 1 at all corners, so attributes
    # are linear in screen space.  model = T(0,0,4) Ry(pi) S(3): frag = (-0.075, -0.15, 4).
    from renderer_amd.types import from_scale
    w, h = 100, 50
    model = from_translation((0, 0, 4.0)) @ from_rotation_y(math.pi) @ from_scale(3.0)
    vs = billboard_scene(2, False, model, w, h, light_pos=(0, 0, 3.0))
    frag = (model.astype(np.float64) @ np.array([0.025, -0.05, 0.0, 1.0]))[:3]
    assert np.allclose(frag, [-0.075, -0.15, 4.0], atol=1e-6)
    want = np.array([0.012, 0.020, 0.036]) + blinn_phong_point(frag, (0, 0, -1), (0, 0, 0), (0, 0, 3.0), 4.0, (1, 1, 1), 2.0)
    got = vs.read_color_hdr()[25, 50]
    assert want[0] > 1.0 and np.allclose(got, want, rtol=2e-3), (got, want)


def test_ka22_billboard_on_faces_the_camera_with_the_models_normal():
    # mode 1: world = (cam.x, cam.y, cam.z) * 0.4 | model.w: a 0.4-unit square facing the camera at the model's
    # origin; rotation and scale of the model are dropped for the position, but the normal stays model * n (:55).
    # 200x100, default camera at the origin, quad centre (0,0,2): corners (+-0.2, +-0.2, 2);
    # x_ndc = +-0.2 * (h/2) / 2, y_ndc = +-0.2 * h / 2 with h = 1/tan(0.65):
    # x_fb = 100 +- 6.5768, y_fb = 50 -+ 6.5768 -> pixel centres 93.5 .. 106.5 and 43.5 .. 56.5 = 14 x 14.
    from renderer_amd.types import from_scale
    w, h = 200, 100
    hh = 1.0 / math.tan(0.65)
    assert abs(0.2 * hh / 2 * 50 - 6.5768) < 1e-3
    model = from_translation((0, 0, 2.0)) @ from_rotation_y(0.3) @ from_scale((3.0, 0.5, 2.0))
    vs = billboard_scene(1, False, model, w, h, light_pos=(0.2, 0.1, 1.0))
    cov = vs.read_coverage()
    want = np.zeros((h, w), np.uint8)
    want[43:57, 93:107] = 1
    assert (cov == want).all()
    r = 100.0 / 99.95
    assert np.allclose(vs.read_depth()[cov == 1], r * (2.0 - 0.05) / 2.0, atol=3e-7)
    # pixel (100, 50): centre (100.5, 50.5) -> view ray through NDC (0.005, -0.01) at z = 2
    frag = np.array([0.005 * 2.0 / (hh / 2.0), -0.01 * 2.0 / hh, 2.0])
    n = (model.astype(np.float64) @ np.array([0, 0, -1.0, 0]))[:3]
    want_c = np.array([0.012, 0.020, 0.036]) + blinn_phong_point(frag, n, (0, 0, 0), (0.2, 0.1, 1.0), 4.0, (1, 1, 1), 2.0)
    got = vs.read_color_hdr()[50, 100]
    assert np.allclose(got, want_c, rtol=3e-3), (got, want_c)
    # turned away (Ry(pi)): the model's normal points from the camera, so only the ambient term is left
    vs = billboard_scene(1, False, from_translation((0, 0, 2.0)) @ from_rotation_y(math.pi), w, h, light_pos=(0.2, 0.1, 1.0))
    assert (vs.read_coverage() == want).all()
    assert np.allclose(vs.read_color_hdr()[50, 100], [0.012, 0.020, 0.036], rtol=2e-3)
