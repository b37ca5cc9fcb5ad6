// TEST INFRASTRUCTURE — CPU restatement (the parity checker), not the product.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/.
//
// CPU restatement of the reference's frame hot path, following these reference files:
//   vertex stage      src/renderer/shaders/render_mesh.wgsl:42-82   (vs_main)
//   ambient/prepass   render_mesh.wgsl:86-111                       (fs_main_ambient_light_depth_prepass)
//   light pass        render_mesh.wgsl:134-181, :256-281            (fs_main_blinn_phong + helpers)
//   shadow lookup     render_mesh.wgsl:184-252, render_shadow_map.wgsl:25-33
//   tone map/present  src/renderer/shaders/show_texture.wgsl:40-61, src/renderer/backend.rs:75-80
//   pass structure    src/renderer/pipeline3d.rs:261-368 (shadow passes, prepass, light passes)
//   fixed function    pipeline3d.rs:454-472 (TriangleList, CCW front, cull back),
//                     :461-467 (Less+write), :507-513 (Equal, no write), :561-575 (additive)
//   targets/clears    src/renderer/visual_server.rs:92-100, :1075-1091, src/color.rs:34
//   host state        visual_server.rs:206-222 (set_camera), :350-474 (lights), :481-506
//                     (instances), :664-750 (mesh/material registration), :771-863 (cascades)
//
// The fixed-function stages (clipping, rasterization, depth test, blending) are NOT in the
// reference: wgpu 0.19 hands them to the GPU driver.  They are restated from the WebGPU /
// Vulkan rules (pixel centre at +0.5, top-left fill rule, clip volume 0<=z<=w, y-down
// framebuffer, 8 sub-pixel bits) — the exact arithmetic is in DESIGN.md "Raster rules".
//
// PARITY UNPINNED: the reference has no tests, fixtures or golden images for this path
//  and cannot be built or run here (Rust + wgpu + window + GPU driver).
// The known-answer tests in tests/ are derived by hand from the WGSL text.
//
// Two formulations are implemented and compared by the tests:
//   mode 0 "deferred": one visibility pass (nearest fragment, ties -> first in draw order),
//                      then one shading evaluation per pixel in f32, one f16 rounding.
//                      This is what the HIP path computes; it is compared bit for bit.
//   mode 1 "forward":  the reference's own pass structure: depth+ambient prepass, then one
//                      additive Equal-depth pass per (mesh, light) into an f16 target.
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstring>
#include <cstdio>
#include <map>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../include/rnd.h"
#include "oracle_math.hpp"

namespace orc {

static constexpr uint32_t TILE = 32;          // sort-first tile edge (must match the product)
static constexpr uint32_t GROUP = 2;          // ownership granularity in tile rows (the HIP library's kOwnerGroup)
static constexpr int SUBPIXEL_BITS = 8;
static constexpr int SUB = 1 << SUBPIXEL_BITS;  // 256
static constexpr float GUARD = 16.0f;         // guard band in NDC units
static constexpr uint32_t SHADOW_DIM = 2048;  // visual_server.rs:366-367

struct Vertex { V3 pos, nrm; float u, v; };
struct Submesh { std::vector<Vertex> verts; std::vector<uint32_t> idx; uint32_t material; };
struct Mesh { std::vector<Submesh> subs; };
struct Material { float base[4]; uint32_t billboard, unlit, texture; };
struct Texture {
    uint32_t w, h, mips;
    std::vector<float> texels;          // all levels, RGBA f32, sRGB-decoded rgb (Rgba8UnormSrgb, backend.rs:376)
    std::vector<size_t> level_offset;   // in texels
    bool opaque;                        // every alpha == 255
};
struct Light {
    uint64_t id; M4 transform; float color[4]; uint32_t kind; float radius;
    M4 cascades[4];                 // world -> light clip of the four shadow cascades, as of set_light
    std::vector<std::vector<float>> shadow;   // per frame: 4 x SHADOW_DIM^2 depths (directional lights)
};
struct Instance { uint64_t id; M4 transform; uint32_t mesh, mat_override; bool casts_shadows; };

struct DrawItem {
    const Submesh* sub;
    M4 model;
    Material mat;
    uint32_t tri_base;
    bool casts_shadows;
    std::vector<V4> clip;     // per-vertex clip position (vs_main)
};

static const uint64_t CLEAR_KEY = (uint64_t(0x3F800000u) << 32) | 0xFFFFFFFFull;

inline uint32_t fbits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float bitsf(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// ------------------------------------------------------------------------------------
// Raster rules (DESIGN.md "Raster rules").  frag(i, j, depth) is called for every covered
// pixel centre of the triangle with clip-space vertices c[0..2].
// ------------------------------------------------------------------------------------
struct ClipVert { float x, y, z, w; };

static inline float plane_dist(const ClipVert& v, int plane) {
    switch (plane) {
        case 0: return v.z;                    // near:  z >= 0
        case 1: return v.w - v.z;              // far:   z <= w
        case 2: return v.x + GUARD * v.w;      // guard band left
        case 3: return GUARD * v.w - v.x;      // guard band right
        case 4: return v.y + GUARD * v.w;      // guard band bottom
        default: return GUARD * v.w - v.y;     // guard band top
    }
}

// intersection always evaluated from the inside vertex towards the outside vertex, so that two
// triangles sharing an edge produce the identical clipped vertex
static inline ClipVert intersect(const ClipVert& in, float din, const ClipVert& out, float dout) {
    float t = din / (din - dout);
    ClipVert r;
    r.x = in.x + t * (out.x - in.x);
    r.y = in.y + t * (out.y - in.y);
    r.z = in.z + t * (out.z - in.z);
    r.w = in.w + t * (out.w - in.w);
    return r;
}

static int clip_polygon(ClipVert* poly, int n) {
    ClipVert tmp[12];
    for (int plane = 0; plane < 6; ++plane) {
        int m = 0;
        for (int k = 0; k < n; ++k) {
            const ClipVert& a = poly[k];
            const ClipVert& b = poly[(k + 1) % n];
            float da = plane_dist(a, plane), db = plane_dist(b, plane);
            bool ia = da >= 0.0f, ib = db >= 0.0f;
            if (ia) tmp[m++] = a;
            if (ia != ib) tmp[m++] = ia ? intersect(a, da, b, db) : intersect(b, db, a, da);
        }
        n = m;
        for (int k = 0; k < n; ++k) poly[k] = tmp[k];
        if (n < 3) return 0;
    }
    return n;
}

struct Snapped { int32_t X, Y; float z; };

// (sub_dx, sub_dy): whole sub-pixel steps added after the snap.  With MSAA, sample k of a pixel at
// offset (ox, oy) from the pixel's corner is the pixel centre of the geometry moved by
// (1/2 - ox, 1/2 - oy): every offset of the 4x pattern is a multiple of 1/16 px, exact on the grid.
static inline bool snap_vertex(const ClipVert& v, float half_w, float half_h, Snapped& s, int32_t sub_dx = 0, int32_t sub_dy = 0) {
    if (!(v.w > 0.0f)) return false;
    float rw = 1.0f / v.w;
    float nx = v.x * rw, ny = v.y * rw, nz = v.z * rw;
    float sx = (nx + 1.0f) * half_w;          // x_fb = (x_ndc + 1) * W/2
    float sy = (1.0f - ny) * half_h;          // y_fb = (1 - y_ndc) * H/2   (y down)
    s.X = (int32_t)std::nearbyint(sx * (float)SUB) + sub_dx;   // round to nearest even, 8 sub-pixel bits
    s.Y = (int32_t)std::nearbyint(sy * (float)SUB) + sub_dy;
    s.z = nz;
    return true;
}

template <class F>
static void raster_snapped(const Snapped& a, const Snapped& b, const Snapped& c, uint32_t W,
                           uint32_t H, F&& frag) {
    // signed double area in y-down framebuffer space; front-facing (CCW with y up) <=> negative
    int64_t A2 = (int64_t)(b.X - a.X) * (c.Y - a.Y) - (int64_t)(c.X - a.X) * (b.Y - a.Y);
    int64_t area2 = -A2;
    if (area2 <= 0) return;                   // back-facing or degenerate: culled
    int32_t xmin = std::min(a.X, std::min(b.X, c.X)), xmax = std::max(a.X, std::max(b.X, c.X));
    int32_t ymin = std::min(a.Y, std::min(b.Y, c.Y)), ymax = std::max(a.Y, std::max(b.Y, c.Y));
    // pixel (i,j) has its centre at (i*256+128, j*256+128); centres inside the closed bbox
    auto ceil_div = [](int64_t n) { return (int32_t)((n >= 0) ? (n + SUB - 1) / SUB : -((-n) / SUB)); };
    auto floor_div = [](int64_t n) { return (int32_t)((n >= 0) ? n / SUB : -((-n + SUB - 1) / SUB)); };
    int32_t i0 = std::max(ceil_div((int64_t)xmin - SUB / 2), 0);
    int32_t i1 = std::min(floor_div((int64_t)xmax - SUB / 2), (int32_t)W - 1);
    int32_t j0 = std::max(ceil_div((int64_t)ymin - SUB / 2), 0);
    int32_t j1 = std::min(floor_div((int64_t)ymax - SUB / 2), (int32_t)H - 1);
    if (i0 > i1 || j0 > j1) return;
    // inside-positive edge functions: w0 opposite a, w1 opposite b, w2 opposite c
    // w0(p) = F(c,b,p), w1(p) = F(a,c,p), w2(p) = F(b,a,p), F(s,t,p) = (t-s) x (p-s)
    const Snapped* S[3] = {&c, &a, &b};
    const Snapped* T[3] = {&b, &c, &a};
    int64_t ex[3], ey[3];
    int bias[3];
    for (int e = 0; e < 3; ++e) {
        ex[e] = (int64_t)T[e]->X - S[e]->X;
        ey[e] = (int64_t)T[e]->Y - S[e]->Y;
        // gradient of w_e: d/dpx = -ey, d/dpy = ex.  left edge: d/dpx > 0; top edge: d/dpx == 0
        // and d/dpy > 0 (interior below, y down).  Non top-left edges exclude w == 0.
        bool top_left = (-ey[e] > 0) || (ey[e] == 0 && ex[e] > 0);
        bias[e] = top_left ? 0 : 1;
    }
    float inv_area = 1.0f / (float)area2;
    float dz1 = b.z - a.z, dz2 = c.z - a.z;
    for (int32_t j = j0; j <= j1; ++j) {
        int64_t py = (int64_t)j * SUB + SUB / 2;
        for (int32_t i = i0; i <= i1; ++i) {
            int64_t px = (int64_t)i * SUB + SUB / 2;
            int64_t w[3];
            bool in = true;
            for (int e = 0; e < 3; ++e) {
                w[e] = ex[e] * (py - S[e]->Y) - ey[e] * (px - S[e]->X);
                if (w[e] < bias[e]) in = false;
            }
            if (!in) continue;
            float l1 = (float)w[1] * inv_area, l2 = (float)w[2] * inv_area;
            float z = (l1 * dz1 + l2 * dz2) + a.z;
            z = std::fmin(std::fmax(z, 0.0f), 1.0f);
            frag((uint32_t)i, (uint32_t)j, z);
        }
    }
}

template <class F>
static void raster_triangle(const V4& c0, const V4& c1, const V4& c2, uint32_t W, uint32_t H,
                            F&& frag, int32_t sub_dx = 0, int32_t sub_dy = 0) {
    ClipVert v[12] = {{c0.x, c0.y, c0.z, c0.w}, {c1.x, c1.y, c1.z, c1.w}, {c2.x, c2.y, c2.z, c2.w}};
    for (int k = 0; k < 3; ++k)
        if (!(std::isfinite(v[k].x) && std::isfinite(v[k].y) && std::isfinite(v[k].z) &&
              std::isfinite(v[k].w)))
            return;
    // trivial reject against the view volume -w<=x<=w, -w<=y<=w, 0<=z<=w
    bool out_l = true, out_r = true, out_b = true, out_t = true, out_n = true, out_f = true;
    bool need_clip = false;
    for (int k = 0; k < 3; ++k) {
        out_l &= v[k].x < -v[k].w; out_r &= v[k].x > v[k].w;
        out_b &= v[k].y < -v[k].w; out_t &= v[k].y > v[k].w;
        out_n &= v[k].z < 0.0f;    out_f &= v[k].z > v[k].w;
        float g = GUARD * v[k].w;
        need_clip |= (v[k].z < 0.0f) | (v[k].z > v[k].w) | (v[k].x < -g) | (v[k].x > g) |
                     (v[k].y < -g) | (v[k].y > g);
    }
    if (out_l | out_r | out_b | out_t | out_n | out_f) return;
    int n = 3;
    if (need_clip) n = clip_polygon(v, 3);
    if (n < 3) return;
    float half_w = 0.5f * (float)W, half_h = 0.5f * (float)H;
    Snapped s[12];
    for (int k = 0; k < n; ++k)
        if (!snap_vertex(v[k], half_w, half_h, s[k], sub_dx, sub_dy)) return;
    for (int k = 1; k + 1 < n; ++k) raster_snapped(s[0], s[k], s[k + 1], W, H, frag);
}

// Perspective-correct barycentrics of pixel (i,j) w.r.t. the ORIGINAL clip-space triangle, by 2D
// homogeneous edge functions in f64 (valid for vertices with w <= 0 too).
// db_dx / db_dy (optional): derivatives of the barycentrics per PIXEL step in x / y, from the same
// affine edge functions (e_i = a_i*xn + b_i*yn + c_i, b_i = e_i / S): d(e_i/S) = (de_i*S - e_i*dS)/S^2.
static inline void barycentrics(const V4& c0, const V4& c1, const V4& c2, uint32_t i, uint32_t j,
                                uint32_t W, uint32_t H, float b[3], float* db_dx = nullptr, float* db_dy = nullptr) {
    double x0 = c0.x, y0 = c0.y, w0 = c0.w, x1 = c1.x, y1 = c1.y, w1 = c1.w, x2 = c2.x, y2 = c2.y,
           w2 = c2.w;
    double xn = ((double)i + 0.5) * (2.0 / (double)W) - 1.0;
    double yn = 1.0 - ((double)j + 0.5) * (2.0 / (double)H);
    // n0 = P1 x P2, n1 = P2 x P0, n2 = P0 x P1 with P = (x, y, w)
    double e0 = ((y1 * w2 - w1 * y2) * xn + (w1 * x2 - x1 * w2) * yn) + (x1 * y2 - y1 * x2);
    double e1 = ((y2 * w0 - w2 * y0) * xn + (w2 * x0 - x2 * w0) * yn) + (x2 * y0 - y2 * x0);
    double e2 = ((y0 * w1 - w0 * y1) * xn + (w0 * x1 - x0 * w1) * yn) + (x0 * y1 - y0 * x1);
    double inv_s = 1.0 / ((e0 + e1) + e2);
    double b0 = e0 * inv_s, b1 = e1 * inv_s, b2 = e2 * inv_s;
    bool ok = std::isfinite(b0) && std::isfinite(b1) && std::isfinite(b2);
    if (!ok) { b0 = 1.0; b1 = 0.0; b2 = 0.0; }
    b[0] = (float)b0; b[1] = (float)b1; b[2] = (float)b2;
    if (db_dx && db_dy) {
        const double ax[3] = {y1 * w2 - w1 * y2, y2 * w0 - w2 * y0, y0 * w1 - w0 * y1};
        const double ay[3] = {w1 * x2 - x1 * w2, w2 * x0 - x2 * w0, w0 * x1 - x0 * w1};
        const double e[3] = {e0, e1, e2};
        const double s = (e0 + e1) + e2, sax = (ax[0] + ax[1]) + ax[2], say = (ay[0] + ay[1]) + ay[2];
        const double step_x = 2.0 / (double)W, step_y = -2.0 / (double)H;
        for (int k = 0; k < 3; ++k) {
            double dx = (ax[k] * s - e[k] * sax) * (inv_s * inv_s) * step_x;
            double dy = (ay[k] * s - e[k] * say) * (inv_s * inv_s) * step_y;
            db_dx[k] = ok && std::isfinite(dx) ? (float)dx : 0.0f;
            db_dy[k] = ok && std::isfinite(dy) ? (float)dy : 0.0f;
        }
    }
}

// ------------------------------------------------------------------------------------
struct Ctx {
    uint32_t W = 0, H = 0;             // resolution of the 3D render target: surface size * size_factor, truncated (visual_server.rs:621-623)
    uint32_t SW = 0, SH = 0;           // surface (render_size)
    float size_factor = 1.0f;
    uint32_t msaa = 1;
    uint32_t tone = RND_TONE_MAPPING_REINHARD;
    int mode = 0;
    int threads = 1;
    uint32_t rank = 0, world = 1;
    std::string err;

    std::unordered_map<uint32_t, Mesh> meshes;
    std::unordered_map<uint32_t, Material> materials;
    std::unordered_map<uint32_t, Texture> textures;
    std::vector<Light> lights;          // insertion order = pass order
    std::vector<Instance> instances;    // insertion order = draw order
    uint32_t next_sprite_material = 0x80000000u;
    std::unordered_map<uint64_t, uint32_t> sprite_material;

    M4 proj = perspective_lh(1.3f, 1.667f, 0.05f, 100.0f);   // Camera::default(), visual_server.rs:64
    M4 view = identity();
    M4 cam = identity();
    M4 inv_pv = identity();            // RenderScene::inv_projection_view (Default = identity), visual_server.rs:898
    float ambient[4] = {0.3f, 0.5f, 0.9f, 0.04f};             // visual_server.rs:67

    // 2D overlay state (Pipeline2d): boxes, glyph quads of every text, font atlas, debug view
    struct Glyph { float pos[2], scale[2]; uint32_t glyph; };      // GlyphInstance, pipeline2d/glyph_instance.rs:5-9
    std::vector<rnd_uibox> uiboxes;
    std::vector<Glyph> glyphs;
    Texture font;                       // Rgba8Unorm (linear), one level
    int fullscreen = 0;                 // 0 none, 1 depth of the 3D target, 2 layer 0 of a light's shadow maps
    uint64_t fullscreen_light = 0;

    // frame outputs
    bool have_frame = false;
    std::vector<uint8_t> overlay;       // the 2D target, SW x SH RGBA8
    std::vector<uint64_t> vis;
    std::vector<float> hdr;
    std::vector<uint8_t> rgba;
    rnd_timings timings{};
    rnd_frame_stats stats{};
};

static Material default_material() { return Material{{1, 1, 1, 1}, 0, 0, RND_NONE}; }

static Mesh quad_mesh() {   // src/mesh.rs:16-37
    Mesh m;
    Submesh s;
    V3 n{0, 0, -1};
    s.verts = {{{-0.5f, 0.5f, 0}, n, 0, 0}, {{0.5f, 0.5f, 0}, n, 1, 0},
               {{-0.5f, -0.5f, 0}, n, 0, 1}, {{0.5f, -0.5f, 0}, n, 1, 1}};
    s.idx = {0, 2, 1, 1, 2, 3};
    s.material = RND_NONE;
    m.subs.push_back(std::move(s));
    return m;
}
static constexpr uint32_t QUAD_MESH_ID = 0xFFFFFFFEu;

static bool owns_tile(const Ctx& c, uint32_t tx, uint32_t ty) {
    if (c.world <= 1) return true;
    (void)tx;
    return c.world - 1u - (ty / GROUP) % c.world == c.rank;   // full-width stripes of GROUP tile rows, dealt round-robin from the last rank
}
static bool owns_pixel(const Ctx& c, uint32_t i, uint32_t j) { return owns_tile(c, i / TILE, j / TILE); }

// vs_main: clip position of one vertex (render_mesh.wgsl:42-82)
static V4 vertex_clip(const M4& pv, const M4& model, const M4& cam, uint32_t billboard, V3 pos) {
    if (billboard == RND_BILLBOARD_ON) {
        M4 t{};
        for (int r = 0; r < 4; ++r) {
            t.c[0 + r] = cam.c[0 + r] * 0.4f;
            t.c[4 + r] = cam.c[4 + r] * 0.4f;
            t.c[8 + r] = cam.c[8 + r] * 0.4f;
            t.c[12 + r] = model.c[12 + r];
        }
        return mul(pv, mul(t, V4{pos.x, pos.y, pos.z, 1.0f}));
    }
    if (billboard == RND_BILLBOARD_FIXED_SIZE) {
        V4 vp = mul(pv, V4{model.c[12], model.c[13], model.c[14], 1.0f});
        V4 q{vp.x / vp.w, vp.y / vp.w, vp.z / vp.w, vp.w / vp.w};
        return V4{q.x + pos.x * 0.4f, q.y + pos.y * 0.4f, q.z, q.w};
    }
    return mul(pv, mul_point(model, pos));
}

struct Frag { V3 pos, nrm; float u, v; float dudx = 0, dvdx = 0, dudy = 0, dvdy = 0; };

// fragment inputs by the deferred formulation: interpolate in object space, transform once
static Frag frag_deferred(const DrawItem& it, const M4& cam, uint32_t t, const float b[3]) {
    const Submesh& s = *it.sub;
    const Vertex& a = s.verts[s.idx[3 * t]], &bb = s.verts[s.idx[3 * t + 1]], &c = s.verts[s.idx[3 * t + 2]];
    V3 p{(b[0] * a.pos.x + b[1] * bb.pos.x) + b[2] * c.pos.x, (b[0] * a.pos.y + b[1] * bb.pos.y) + b[2] * c.pos.y,
         (b[0] * a.pos.z + b[1] * bb.pos.z) + b[2] * c.pos.z};
    V3 n{(b[0] * a.nrm.x + b[1] * bb.nrm.x) + b[2] * c.nrm.x, (b[0] * a.nrm.y + b[1] * bb.nrm.y) + b[2] * c.nrm.y,
         (b[0] * a.nrm.z + b[1] * bb.nrm.z) + b[2] * c.nrm.z};
    Frag f;
    if (it.mat.billboard == RND_BILLBOARD_ON) {
        M4 tm{};
        for (int r = 0; r < 4; ++r) {
            tm.c[0 + r] = cam.c[0 + r] * 0.4f; tm.c[4 + r] = cam.c[4 + r] * 0.4f;
            tm.c[8 + r] = cam.c[8 + r] * 0.4f; tm.c[12 + r] = it.model.c[12 + r];
        }
        V4 w = mul(tm, V4{p.x, p.y, p.z, 1.0f});
        f.pos = {w.x, w.y, w.z};
        f.nrm = mul_dir(it.model, n);
    } else {
        V4 w = mul_point(it.model, p);
        f.pos = {w.x, w.y, w.z};
        f.nrm = (it.mat.billboard == RND_BILLBOARD_FIXED_SIZE) ? n : mul_dir(it.model, n);
    }
    f.u = (b[0] * a.u + b[1] * bb.u) + b[2] * c.u;
    f.v = (b[0] * a.v + b[1] * bb.v) + b[2] * c.v;
    return f;
}

// fragment inputs by the reference's formulation: vs_main outputs per vertex, interpolated
static Frag frag_forward(const DrawItem& it, const M4& cam, uint32_t t, const float b[3]) {
    const Submesh& s = *it.sub;
    V3 wp[3], wn[3]; float uu[3], vv[3];
    for (int k = 0; k < 3; ++k) {
        const Vertex& vx = s.verts[s.idx[3 * t + k]];
        V4 w = mul_point(it.model, vx.pos);
        wp[k] = {w.x, w.y, w.z};
        wn[k] = mul_dir(it.model, vx.nrm);
        if (it.mat.billboard == RND_BILLBOARD_ON) {
            M4 tm{};
            for (int r = 0; r < 4; ++r) {
                tm.c[0 + r] = cam.c[0 + r] * 0.4f; tm.c[4 + r] = cam.c[4 + r] * 0.4f;
                tm.c[8 + r] = cam.c[8 + r] * 0.4f; tm.c[12 + r] = it.model.c[12 + r];
            }
            V4 w2 = mul(tm, V4{vx.pos.x, vx.pos.y, vx.pos.z, 1.0f});
            wp[k] = {w2.x, w2.y, w2.z};
        } else if (it.mat.billboard == RND_BILLBOARD_FIXED_SIZE) {
            wn[k] = vx.nrm;
        }
        uu[k] = vx.u; vv[k] = vx.v;
    }
    Frag f;
    f.pos = {(b[0] * wp[0].x + b[1] * wp[1].x) + b[2] * wp[2].x, (b[0] * wp[0].y + b[1] * wp[1].y) + b[2] * wp[2].y,
             (b[0] * wp[0].z + b[1] * wp[1].z) + b[2] * wp[2].z};
    f.nrm = {(b[0] * wn[0].x + b[1] * wn[1].x) + b[2] * wn[2].x, (b[0] * wn[0].y + b[1] * wn[1].y) + b[2] * wn[2].y,
             (b[0] * wn[0].z + b[1] * wn[1].z) + b[2] * wn[2].z};
    f.u = (b[0] * uu[0] + b[1] * uu[1]) + b[2] * uu[2];
    f.v = (b[0] * vv[0] + b[1] * vv[1]) + b[2] * vv[2];
    return f;
}

// base = material.base_color * textureSample(...) (render_mesh.wgsl:89, :142); untextured
// materials bind the 1x1 white texture (visual_server.rs:736).
// textureSample with the reference's "filtered" sampler (backend.rs:414-426): Repeat addressing,
// linear mag/min/mip.  Restated from the WebGPU/Vulkan rules, isotropic (anisotropy 16 is
// implementation-defined and not modelled): rho = the longer of the two pixel footprints in texels,
// lambda = log2(rho) clamped to [0, mips-1]; bilinear taps at floor(x - 0.5), weights
// a*(1-t) + b*t; two levels blended by frac(lambda).
static void bilinear(const Texture& t, uint32_t level, float u, float v, float out[4]) {
    const uint32_t wl = std::max(1u, t.w >> level), hl = std::max(1u, t.h >> level);
    const float x = u * (float)wl - 0.5f, y = v * (float)hl - 0.5f;
    const float xf = std::floor(x), yf = std::floor(y);
    const float fx = x - xf, fy = y - yf;
    auto wrap = [](float f, uint32_t n) {
        long long k = (long long)f % (long long)n;
        return (uint32_t)(k < 0 ? k + n : k);
    };
    const uint32_t x0 = wrap(xf, wl), x1 = wrap(xf + 1.0f, wl), y0 = wrap(yf, hl), y1 = wrap(yf + 1.0f, hl);
    const float* base = t.texels.data() + 4 * t.level_offset[level];
    const float* t00 = base + 4 * ((size_t)y0 * wl + x0), *t10 = base + 4 * ((size_t)y0 * wl + x1);
    const float* t01 = base + 4 * ((size_t)y1 * wl + x0), *t11 = base + 4 * ((size_t)y1 * wl + x1);
    for (int k = 0; k < 4; ++k) {
        float top = t00[k] * (1.0f - fx) + t10[k] * fx, bot = t01[k] * (1.0f - fx) + t11[k] * fx;
        out[k] = top * (1.0f - fy) + bot * fy;
    }
}
static void sample_texture(const Texture& t, const Frag& f, float out[4]) {
    const float ax = f.dudx * (float)t.w, ay = f.dvdx * (float)t.h, bx = f.dudy * (float)t.w, by = f.dvdy * (float)t.h;
    const float rho2 = std::fmax(ax * ax + ay * ay, bx * bx + by * by);
    float lambda = 0.5f * std::log2(rho2);                     // log2(sqrt(rho2)); -inf for a constant uv
    lambda = std::isnan(lambda) ? 0.0f : clampf(lambda, 0.0f, (float)(t.mips - 1));
    const float lf = std::floor(lambda), frac = lambda - lf;
    const uint32_t l0 = (uint32_t)lf, l1 = std::min(l0 + 1, t.mips - 1);
    float a[4], b[4];
    bilinear(t, l0, f.u, f.v, a);
    if (frac > 0.0f) {
        bilinear(t, l1, f.u, f.v, b);
        for (int k = 0; k < 4; ++k) out[k] = a[k] * (1.0f - frac) + b[k] * frac;
    } else {
        for (int k = 0; k < 4; ++k) out[k] = a[k];
    }
}
static void base_color(const Ctx& c, const Material& m, const Frag* f, float out[4]) {
    float texel[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    if (m.texture != RND_NONE && f) {
        auto it = c.textures.find(m.texture);
        if (it != c.textures.end()) sample_texture(it->second, *f, texel);
    }
    for (int k = 0; k < 4; ++k) out[k] = m.base[k] * texel[k];
}

// compute_light_blinn_phong (render_mesh.wgsl:256-276) with shininess 8
static V3 blinn_phong(V3 base, V3 n, V3 to_view, V3 light_dir, V3 light_color, float intensity) {
    V3 to_light = neg(light_dir);
    float diff_i = std::fmax(dot(n, to_light), 0.0f);
    V3 diffuse = (light_color * intensity) * diff_i;
    V3 h = normalize(to_light + to_view);
    float s = std::fmax(dot(n, h), 0.0f);
    float s2 = s * s, s4 = s2 * s2, s8 = s4 * s4;       // pow(x, 8.0)
    V3 spec = (light_color * intensity) * s8;
    return base * (diffuse + spec);
}

// compute_shadow_cascade_projviews (visual_server.rs:771-863): slices of the camera frustum
// (ratios 0-0.05-0.1-0.3-1, :88), a light view looking along light_dir from each slice's
// centroid, an orthographic box around the slice with its z range widened tenfold.
static void shadow_cascade_projviews(const M4& inv_pv, V3 light_dir, M4 out[4]) {
    auto frustum_point = [&](float x, float y, float z) {
        V4 fp = mul(inv_pv, V4{x, y, z, 1.0f});
        return V4{fp.x / fp.w, fp.y / fp.w, fp.z / fp.w, fp.w / fp.w};
    };
    const V4 n[4] = {frustum_point(-1, 1, 0), frustum_point(1, 1, 0), frustum_point(-1, -1, 0), frustum_point(1, -1, 0)};
    const V4 f[4] = {frustum_point(-1, 1, 1), frustum_point(1, 1, 1), frustum_point(-1, -1, 1), frustum_point(1, -1, 1)};
    V4 edge[4];
    for (int k = 0; k < 4; ++k) edge[k] = V4{f[k].x - n[k].x, f[k].y - n[k].y, f[k].z - n[k].z, f[k].w - n[k].w};
    const float ratios[4][2] = {{0.0f, 0.05f}, {0.05f, 0.1f}, {0.1f, 0.3f}, {0.3f, 1.0f}};
    for (int cascade = 0; cascade < 4; ++cascade) {
        V4 corner[8];
        for (int k = 0; k < 4; ++k) {
            const float rn = ratios[cascade][0], rf = ratios[cascade][1];
            corner[k] = V4{n[k].x + edge[k].x * rn, n[k].y + edge[k].y * rn, n[k].z + edge[k].z * rn, n[k].w + edge[k].w * rn};
            corner[4 + k] = V4{n[k].x + edge[k].x * rf, n[k].y + edge[k].y * rf, n[k].z + edge[k].z * rf, n[k].w + edge[k].w * rf};
        }
        V4 sum = corner[0];
        for (int k = 1; k < 8; ++k) sum = V4{sum.x + corner[k].x, sum.y + corner[k].y, sum.z + corner[k].z, sum.w + corner[k].w};
        V3 center{sum.x / 8.0f, sum.y / 8.0f, sum.z / 8.0f};
        M4 view = look_to_lh(center, light_dir, V3{0, 1, 0});
        float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};   // f32::MAX / f32::MIN
        for (int k = 0; k < 8; ++k) {
            V4 p = mul(view, corner[k]);
            mn[0] = std::fmin(mn[0], p.x); mx[0] = std::fmax(mx[0], p.x);
            mn[1] = std::fmin(mn[1], p.y); mx[1] = std::fmax(mx[1], p.y);
            mn[2] = std::fmin(mn[2], p.z); mx[2] = std::fmax(mx[2], p.z);
        }
        const float z_mult = 10.0f;
        if (mn[2] < 0.0f) mn[2] *= z_mult; else mn[2] /= z_mult;
        if (mx[2] < 0.0f) mx[2] /= z_mult; else mx[2] *= z_mult;
        out[cascade] = mul(orthographic_lh(mn[0], mx[0], mn[1], mx[1], mn[2], mx[2]), view);
    }
}

// compute_light_occlusion (render_mesh.wgsl:184-252).  The shader selects layer 3 beyond view
// depth 30 but LightUniform holds three matrices (:118, visual_server.rs:409-413); naga's
// "Restrict" index policy clamps the index, so layer 3 is looked up through cascade 2's matrix.
// Nearest sampler, ClampToBorder with OpaqueWhite (backend.rs:441-453): depth 1.0 outside.
static float light_occlusion(const Ctx& c, const Light& l, V3 frag_pos, V3 normal, V3 light_dir) {
    if (dot(normal, light_dir) > 0.0f) return 0.0f;
    int layer; float bias_mod;
    const float view_depth = mul_point(c.view, frag_pos).z;
    if (view_depth <= 0.05f * 100.0f) { layer = 0; bias_mod = 0.03f; }
    else if (view_depth <= 0.1f * 100.0f) { layer = 1; bias_mod = 0.05f; }
    else if (view_depth <= 0.3f * 100.0f) { layer = 2; bias_mod = 0.1f; }
    else { layer = 3; bias_mod = 0.2f; }
    V3 depth_offset = neg(light_dir) * 0.3f;
    V3 normal_offset = normal * 0.8f;
    V3 biased = frag_pos + (depth_offset + normal_offset) * bias_mod;
    V4 ls = mul(l.cascades[layer < 2 ? layer : 2], V4{biased.x, biased.y, biased.z, 1.0f});
    float nx = ls.x / ls.w, ny = ls.y / ls.w, nz = ls.z / ls.w;
    float u = (nx + 1.0f) * 0.5f, v = 1.0f - (ny + 1.0f) * 0.5f;
    float frag_depth = clampf(nz, 0.0f, 1.0f);
    const float texel = 1.0f / (float)SHADOW_DIM;
    const std::vector<float>& map = l.shadow[layer];
    float occlusion = 0.0f, samples = 0.0f;
    for (int x = -1; x <= 1; ++x)
        for (int y = -1; y <= 1; ++y) {
            samples += 1.0f;
            float su = u + (float)x * texel, sv = v + (float)y * texel;
            float occluder = 1.0f;                                  // border colour
            if (su >= 0.0f && su < 1.0f && sv >= 0.0f && sv < 1.0f) {
                uint32_t ix = std::min((uint32_t)std::floor(su * (float)SHADOW_DIM), SHADOW_DIM - 1);
                uint32_t iy = std::min((uint32_t)std::floor(sv * (float)SHADOW_DIM), SHADOW_DIM - 1);
                occluder = map[(size_t)iy * SHADOW_DIM + ix];
            }
            if (frag_depth > occluder) occlusion += 1.0f;
        }
    return occlusion / samples;
}

// contribution of one light to one fragment (fs_main_blinn_phong, render_mesh.wgsl:134-181);
// returns false when the fragment is discarded for this light
static bool light_contribution(const Ctx& c, const Light& l, const Frag& f, V3 base, V3& out) {
    V3 n = normalize(f.nrm);
    V3 cam_pos{c.cam.c[12], c.cam.c[13], c.cam.c[14]};
    V3 to_view = normalize(cam_pos - f.pos);
    if (l.kind == RND_LIGHT_DIRECTIONAL) {
        V3 dir{l.transform.c[8], l.transform.c[9], l.transform.c[10]};
        float occlusion = light_occlusion(c, l, f.pos, n, dir);
        out = blinn_phong(base, n, to_view, dir, V3{l.color[0], l.color[1], l.color[2]},
                          l.color[3] * (1.0f - occlusion));
        return true;
    }
    if (l.kind == RND_LIGHT_POINT) {
        V3 lp{l.transform.c[12], l.transform.c[13], l.transform.c[14]};
        V3 d = f.pos - lp;
        float dist = std::sqrt(dot(d, d));
        if (dist > l.radius) return false;
        V3 dir = d * (1.0f / dist);                      // normalize(frag_pos - light_pos)
        float lin = clampf((l.radius - dist) / l.radius, 0.0f, 1.0f);
        float att = (lin * lin) * (3.0f - 2.0f * lin);   // smoothstep(0,1,lin)
        out = blinn_phong(base, n, to_view, dir, V3{l.color[0], l.color[1], l.color[2]},
                          l.color[3] * att);
        return true;
    }
    out = {0, 0, 0};
    return true;
}

// show_texture.wgsl:40-61 + sRGB surface
static void present_pixel(const Ctx& c, const float hdr[3], uint8_t out[4]) {
    float r = hdr[0], g = hdr[1], b = hdr[2];
    if (c.tone == RND_TONE_MAPPING_REINHARD) {
        float lum = (0.2126f * r + 0.7152f * g) + 0.0722f * b;
        float d = lum + 1.0f;
        r = r / d; g = g / d; b = b / d;
    }
    out[0] = srgb8(r); out[1] = srgb8(g); out[2] = srgb8(b); out[3] = 255;
}

static void atomic_min(std::atomic<uint64_t>& a, uint64_t v) {
    uint64_t cur = a.load(std::memory_order_relaxed);
    while (v < cur && !a.compare_exchange_weak(cur, v, std::memory_order_relaxed)) {}
}

template <class F>
static void parallel_for(int threads, size_t n, F&& f) {
    if (threads <= 1 || n < 2) { f(0, n); return; }
    std::vector<std::thread> th;
    size_t chunk = (n + threads - 1) / threads;
    for (int t = 0; t < threads; ++t) {
        size_t lo = std::min(n, t * chunk), hi = std::min(n, lo + chunk);
        if (lo < hi) th.emplace_back([=, &f] { f(lo, hi); });
    }
    for (auto& x : th) x.join();
}

static int build_items(Ctx& c, std::vector<DrawItem>& items) {
    uint32_t tri_base = 0;
    for (const Instance& inst : c.instances) {
        auto mit = c.meshes.find(inst.mesh);
        if (mit == c.meshes.end()) { c.err = "mesh id not registered"; return RND_ERR_UNKNOWN_ID; }
        for (const Submesh& s : mit->second.subs) {
            uint32_t mid = inst.mat_override != RND_NONE ? inst.mat_override : s.material;
            Material mat = default_material();
            if (mid != RND_NONE) {
                auto it = c.materials.find(mid);
                if (it == c.materials.end()) { c.err = "material id not registered"; return RND_ERR_UNKNOWN_ID; }
                mat = it->second;
            }
            DrawItem d;
            d.sub = &s; d.model = inst.transform; d.mat = mat; d.tri_base = tri_base;
            d.casts_shadows = inst.casts_shadows;
            tri_base += (uint32_t)(s.idx.size() / 3);
            items.push_back(std::move(d));
        }
    }
    return RND_OK;
}

// ------------------------------------------------------------------------------------
// 2D overlay (DESIGN.md raster rule 14).  Pipeline2d::render (pipeline2d.rs:266-310) draws into a
// surface-sized Rgba8UnormSrgb target cleared to transparent: the full-screen debug view, the UI
// boxes, the texts, all with BlendState::ALPHA_BLENDING (rgb = s.rgb*s.a + d.rgb*(1-s.a),
// a = s.a + d.a*(1-s.a)); Backend::render (backend.rs:470-491) then draws that target over the
// presented 3D image with the same blend state.  A fixed-point target clamps the source to [0,1]
// before blending and stores 8 bits after every draw, so the destination is re-decoded each time.
// Quads are instanced 4-vertex strips, axis-aligned: the two triangles share the diagonal and the
// top-left rule gives every pixel centre in [left, right) x [top, bottom) to exactly one of them.
// ------------------------------------------------------------------------------------
static const float* srgb_decode_lut() {
    static float lut[256];
    static bool init = false;
    if (!init) {
        for (int v = 0; v < 256; ++v) {
            double x = v / 255.0;
            lut[v] = (float)(x <= 0.04045 ? x / 12.92 : std::pow((x + 0.055) / 1.055, 2.4));
        }
        init = true;
    }
    return lut;
}

static inline float sat(float x) { return std::fmin(std::fmax(x, 0.0f), 1.0f); }     // NaN -> 0
static inline float smoothstep(float e0, float e1, float x) {
    float t = sat((x - e0) / (e1 - e0));
    return (t * t) * (3.0f - 2.0f * t);
}

// ALPHA_BLENDING of src (linear rgb, alpha) onto an Rgba8UnormSrgb texel
static inline void blend_srgb8(uint8_t d[4], const float src[4]) {
    const float* lut = srgb_decode_lut();
    const float a = sat(src[3]);
    for (int k = 0; k < 3; ++k) d[k] = srgb8(sat(src[k]) * a + lut[d[k]] * (1.0f - a));
    const float oa = a * 1.0f + ((float)d[3] / 255.0f) * (1.0f - a);
    d[3] = (uint8_t)(int)(sat(oa) * 255.0f + 0.5f);
}

// An instanced quad after the vertex stage (uibox.wgsl:30-42, text.wgsl:39-51): the corner
// (cx, cy) in {0,1}^2 sits at pos + c*size with y measured from the top, goes to clip space and
// through the viewport transform, and is snapped like any other vertex (rule 2).
struct Quad2D { int32_t xa, xb, yt, yb; };
static Quad2D quad_2d(const float pos[2], const float size[2], uint32_t SW, uint32_t SH) {
    const float vw = (float)SW, vh = (float)SH, half_w = 0.5f * vw, half_h = 0.5f * vh;
    auto snap_x = [&](float cx) {
        float tp = cx * size[0] + pos[0];
        float clip = (tp / vw) * 2.0f - 1.0f;
        float sx = (clip + 1.0f) * half_w;
        return (int32_t)std::fmin(std::fmax(std::nearbyint(sx * (float)SUB), -1.0e9f), 1.0e9f);
    };
    auto snap_y = [&](float cy) {
        float tp = cy * size[1] + ((vh - pos[1]) - size[1]);
        float clip = (tp / vh) * 2.0f - 1.0f;
        float sy = (1.0f - clip) * half_h;
        return (int32_t)std::fmin(std::fmax(std::nearbyint(sy * (float)SUB), -1.0e9f), 1.0e9f);
    };
    return Quad2D{snap_x(0.0f), snap_x(1.0f), snap_y(1.0f), snap_y(0.0f)};
}
// front-facing (counter-clockwise, cull Back: pipeline2d.rs:465-470) and covering the centre of (i, j);
// u, v = the quad's own 0..1 coordinates there (v = 0 at the top: text.wgsl:54-57)
static inline bool quad_covers(const Quad2D& q, uint32_t i, uint32_t j, float& u, float& v) {
    const int64_t area = (int64_t)(q.xb - q.xa) * (int64_t)(q.yb - q.yt);
    if (area <= 0) return false;
    const int32_t px = (int32_t)i * SUB + SUB / 2, py = (int32_t)j * SUB + SUB / 2;
    if (px < std::min(q.xa, q.xb) || px >= std::max(q.xa, q.xb) || py < std::min(q.yt, q.yb) || py >= std::max(q.yt, q.yb)) return false;
    u = (float)(px - q.xa) / (float)(q.xb - q.xa);
    v = (float)(py - q.yt) / (float)(q.yb - q.yt);
    return true;
}

static void render_overlay(Ctx& c) {
    const uint32_t SW = c.SW, SH = c.SH;
    c.overlay.assign((size_t)SW * SH * 4, 0);
    if (!c.fullscreen && c.uiboxes.empty() && c.glyphs.empty()) return;
    struct Box { Quad2D q; float color[4]; };
    std::vector<Box> boxes;
    for (const rnd_uibox& b : c.uiboxes) {
        Box o; o.q = quad_2d(b.position, b.size, SW, SH);
        for (int k = 0; k < 3; ++k) o.color[k] = std::pow(b.color[k], 2.2f);      // uibox.wgsl:27-28
        o.color[3] = b.color[3];
        boxes.push_back(o);
    }
    std::vector<Quad2D> gq;
    for (const Ctx::Glyph& g : c.glyphs) gq.push_back(quad_2d(g.pos, g.scale, SW, SH));
    // the debug view's source: depth of the 3D target (sample 0) or layer 0 of a light's cascade maps
    const std::vector<float>* shadow0 = nullptr;
    if (c.fullscreen == 2)
        for (const Light& l : c.lights)
            if (l.id == c.fullscreen_light && !l.shadow.empty()) shadow0 = &l.shadow[0];
    const float* lut = srgb_decode_lut();
    parallel_for(c.threads, SH, [&](size_t lo, size_t hi) {
        for (size_t j = lo; j < hi; ++j)
            for (uint32_t i = 0; i < SW; ++i) {
                if (c.world > 1 && !owns_pixel(c, i, (uint32_t)j)) continue;
                uint8_t* d = &c.overlay[((size_t)j * SW + i) * 4];
                if (c.fullscreen) {
                    // fullscreen_texture.wgsl: uv of the pixel centre, nearest + ClampToEdge sampler
                    const float u = ((float)i + 0.5f) / (float)SW, v = ((float)j + 0.5f) / (float)SH;
                    float depth = 1.0f;
                    if (c.fullscreen == 1) {
                        uint32_t x = std::min((uint32_t)std::floor(u * (float)c.W), c.W - 1), y = std::min((uint32_t)std::floor(v * (float)c.H), c.H - 1);
                        depth = bitsf((uint32_t)(c.vis[(size_t)y * c.W + x] >> 32));
                    } else if (shadow0) {
                        uint32_t x = std::min((uint32_t)std::floor(u * (float)SHADOW_DIM), SHADOW_DIM - 1), y = std::min((uint32_t)std::floor(v * (float)SHADOW_DIM), SHADOW_DIM - 1);
                        depth = (*shadow0)[(size_t)y * SHADOW_DIM + x];
                    }
                    const float dd = (2.0f * 0.5f) / ((50.0f + 0.5f) - depth * (50.0f - 0.5f));     // linearize_depth :37-42
                    const float src[4] = {dd, dd, dd, 1.0f};
                    blend_srgb8(d, src);
                }
                float u, v;
                for (const Box& b : boxes)
                    if (quad_covers(b.q, i, (uint32_t)j, u, v)) blend_srgb8(d, b.color);
                for (size_t k = 0; k < gq.size(); ++k) {
                    if (!quad_covers(gq[k], i, (uint32_t)j, u, v)) continue;
                    const Ctx::Glyph& g = c.glyphs[k];
                    // text.wgsl:58-65: the glyph's cell of the atlas; :76-105: signed distance field
                    const float cell = 1.0f / 128.0f;
                    const float u0 = (0.0f / 128.0f) + (float)g.glyph * cell, u1 = (1.0f / 128.0f) + (float)g.glyph * cell;
                    const float tu = u0 + (u1 - u0) * u;
                    float sd[4];
                    bilinear(c.font, 0, tu, v, sd);
                    const float ratio = (g.scale[1] - 12.0f) / (24.0f - 12.0f);
                    const float smoothing = std::fmin(std::fmax(0.25f + ratio * (0.15f - 0.25f), 0.1f), 0.9f);
                    const float fill_dist = smoothstep(0.37f - smoothing, 0.37f + smoothing, sd[2]);
                    const float outline_dist = smoothstep(0.8f - smoothing, 0.8f + smoothing, sd[2]);
                    const float fill_alpha = 1.0f - fill_dist, alpha = 1.0f - outline_dist;
                    const float col = 0.0f * (1.0f - fill_alpha) + 1.0f * fill_alpha;              // mix(outline, fill, fill_alpha)
                    const float src[4] = {col, col, col, alpha};
                    blend_srgb8(d, src);
                }
                // Backend::render: the 2D target over the presented 3D image (filtered sampler at the
                // surface's own resolution = the texel itself), ALPHA_BLENDING onto the sRGB surface
                if (d[3] == 0) continue;        // s.a = 0 leaves the surface texel as it is
                uint8_t* f = &c.rgba[((size_t)j * SW + i) * 4];
                const float a = (float)d[3] / 255.0f;
                for (int k = 0; k < 3; ++k) f[k] = srgb8(lut[d[k]] * a + lut[f[k]] * (1.0f - a));
            }
    });
}

static int render(Ctx& c) {
    auto t_begin = std::chrono::steady_clock::now();
    const uint32_t W = c.W, H = c.H;
    const size_t P = (size_t)W * H;
    std::vector<DrawItem> items;
    int rc = build_items(c, items);
    if (rc != RND_OK) return rc;
    M4 pv = mul(c.proj, c.view);    // render_mesh.wgsl:48

    // K1: vertex stage
    uint64_t nv = 0, nt = 0;
    for (DrawItem& it : items) {
        const Submesh& s = *it.sub;
        it.clip.resize(s.verts.size());
        parallel_for(c.threads, s.verts.size(), [&](size_t lo, size_t hi) {
            for (size_t k = lo; k < hi; ++k)
                it.clip[k] = vertex_clip(pv, it.model, c.cam, it.mat.billboard, s.verts[k].pos);
        });
        nv += s.verts.size();
        nt += s.idx.size() / 3;
    }

    // Shadow maps of the directional lights: depth-only passes over every item that casts shadows
    // (pipeline3d.rs:268-313; no fragment stage, so no alpha test; render_shadow_map.wgsl:25-33 uses
    // the model matrix only, billboard modes do not apply), cull Back, depth Less against 1.0.
    // The reference also fills maps for point lights, which nothing reads.
    for (Light& l : c.lights) {
        l.shadow.clear();
        if (l.kind != RND_LIGHT_DIRECTIONAL) continue;
        l.shadow.resize(4);
        for (int cascade = 0; cascade < 4; ++cascade) {
            const size_t n_px = (size_t)SHADOW_DIM * SHADOW_DIM;
            std::vector<std::atomic<uint32_t>> depth(n_px);
            for (size_t k = 0; k < n_px; ++k) depth[k].store(0x3F800000u, std::memory_order_relaxed);
            for (const DrawItem& it : items) {
                if (!it.casts_shadows) continue;
                const Submesh& s = *it.sub;
                std::vector<V4> lc(s.verts.size());
                parallel_for(c.threads, s.verts.size(), [&](size_t lo, size_t hi) {
                    for (size_t k = lo; k < hi; ++k) lc[k] = mul(l.cascades[cascade], mul_point(it.model, s.verts[k].pos));
                });
                parallel_for(c.threads, s.idx.size() / 3, [&](size_t lo, size_t hi) {
                    for (size_t t = lo; t < hi; ++t)
                        raster_triangle(lc[s.idx[3 * t]], lc[s.idx[3 * t + 1]], lc[s.idx[3 * t + 2]], SHADOW_DIM, SHADOW_DIM,
                                        [&](uint32_t i, uint32_t j, float z) {
                                            if (!(z < 1.0f)) return;
                                            std::atomic<uint32_t>& a = depth[(size_t)j * SHADOW_DIM + i];
                                            uint32_t v = fbits(z), cur = a.load(std::memory_order_relaxed);
                                            while (v < cur && !a.compare_exchange_weak(cur, v, std::memory_order_relaxed)) {}
                                        });
                });
            }
            l.shadow[cascade].resize(n_px);
            for (size_t k = 0; k < n_px; ++k) {
                uint32_t bits = depth[k].load(std::memory_order_relaxed);
                std::memcpy(&l.shadow[cascade][k], &bits, 4);
            }
        }
    }

    // K2+K3: visibility (nearest fragment; ties -> lowest triangle id = first in draw order).
    // Materials whose alpha test fails for every fragment (untextured, base.a < 0.5) draw nothing
    // (render_mesh.wgsl:91-92, :144-146).
    // MSAA (sample_count 4, visual_server.rs:174-181): one visibility plane per sample, the standard 4x
    // pattern (offsets from the pixel corner in 1/16 px); coverage and depth are per sample, the
    // fragment stage (alpha test, shading) runs once per pixel at its centre.
    const uint32_t S = c.msaa;
    static const int32_t kSampleOffset16[4][2] = {{6, 2}, {14, 6}, {2, 10}, {10, 14}};
    std::vector<std::atomic<uint64_t>> vis(P * S);
    for (size_t k = 0; k < P * S; ++k) vis[k].store(CLEAR_KEY, std::memory_order_relaxed);
    for (uint32_t smp = 0; smp < S; ++smp) {
    const int32_t sub_dx = S == 1 ? 0 : SUB / 2 - kSampleOffset16[smp][0] * (SUB / 16);
    const int32_t sub_dy = S == 1 ? 0 : SUB / 2 - kSampleOffset16[smp][1] * (SUB / 16);
    for (const DrawItem& it : items) {
        float bc[4];
        base_color(c, it.mat, nullptr, bc);     // texel alpha <= 1: below 0.5 here means below 0.5 for every fragment
        if (bc[3] < 0.5f) continue;
        const Submesh& s = *it.sub;
        size_t ntri = s.idx.size() / 3;
        // textures with alpha: base.a < 0.5 is decided per fragment (render_mesh.wgsl:89-92)
        const Texture* alpha_tex = nullptr;
        if (it.mat.texture != RND_NONE) {
            auto tx = c.textures.find(it.mat.texture);
            if (tx != c.textures.end() && !tx->second.opaque) alpha_tex = &tx->second;
        }
        parallel_for(c.threads, ntri, [&](size_t lo, size_t hi) {
            for (size_t t = lo; t < hi; ++t) {
                uint32_t id = it.tri_base + (uint32_t)t;
                raster_triangle(it.clip[s.idx[3 * t]], it.clip[s.idx[3 * t + 1]], it.clip[s.idx[3 * t + 2]], W, H,
                                [&](uint32_t i, uint32_t j, float z) {
                                    if (!(z < 1.0f)) return;            // depth Less against clear 1.0
                                    if (!owns_pixel(c, i, j)) return;
                                    if (alpha_tex) {
                                        const Vertex& va = s.verts[s.idx[3 * t]], &vb = s.verts[s.idx[3 * t + 1]], &vc = s.verts[s.idx[3 * t + 2]];
                                        float b[3], dbx[3], dby[3];
                                        barycentrics(it.clip[s.idx[3 * t]], it.clip[s.idx[3 * t + 1]], it.clip[s.idx[3 * t + 2]], i, j, W, H, b, dbx, dby);
                                        Frag f{};
                                        f.u = (b[0] * va.u + b[1] * vb.u) + b[2] * vc.u; f.v = (b[0] * va.v + b[1] * vb.v) + b[2] * vc.v;
                                        f.dudx = (dbx[0] * va.u + dbx[1] * vb.u) + dbx[2] * vc.u; f.dvdx = (dbx[0] * va.v + dbx[1] * vb.v) + dbx[2] * vc.v;
                                        f.dudy = (dby[0] * va.u + dby[1] * vb.u) + dby[2] * vc.u; f.dvdy = (dby[0] * va.v + dby[1] * vb.v) + dby[2] * vc.v;
                                        float texel[4];
                                        sample_texture(*alpha_tex, f, texel);
                                        if (it.mat.base[3] * texel[3] < 0.5f) return;
                                    }
                                    atomic_min(vis[(size_t)smp * P + (size_t)j * W + i], ((uint64_t)fbits(z) << 32) | id);
                                }, sub_dx, sub_dy);
            }
        });
    }
    }
    c.vis.resize(P * S);        // read-backs show sample 0
    for (size_t k = 0; k < P * S; ++k) c.vis[k] = vis[k].load(std::memory_order_relaxed);

    // K4: shading
    c.hdr.assign(P * 3, 0.0f);
    c.rgba.assign(P * 4, 0);
    const float clear[3] = {round_f16(0.02f), round_f16(0.04f), round_f16(0.08f)};   // Color::GRUE
    std::vector<uint32_t> tri_bases;
    for (const DrawItem& it : items) tri_bases.push_back(it.tri_base);
    auto item_of = [&](uint32_t id) {
        size_t k = std::upper_bound(tri_bases.begin(), tri_bases.end(), id) - tri_bases.begin();
        return k - 1;
    };
    if (c.mode != 0 && S != 1) { c.err = "the forward restatement covers sample count 1 only"; return RND_ERR_UNSUPPORTED; }
    if (c.mode != 0)
        for (const DrawItem& it : items)
            if (it.mat.texture != RND_NONE) { c.err = "the forward restatement covers untextured materials only"; return RND_ERR_UNSUPPORTED; }
    if (c.mode == 0) {
        // colour of the fragment of triangle `id` at pixel (i, j), before the f16 store
        auto shade = [&](uint32_t id, uint32_t i, uint32_t j) {
            const DrawItem& it = items[item_of(id)];
            uint32_t t = id - it.tri_base;
            const Submesh& s = *it.sub;
            float b[3], dbx[3], dby[3];
            barycentrics(it.clip[s.idx[3 * t]], it.clip[s.idx[3 * t + 1]], it.clip[s.idx[3 * t + 2]], i, j, W, H, b, dbx, dby);
            Frag f = frag_deferred(it, c.cam, t, b);
            {   // screen-space uv derivatives for the mip level, analytic from the barycentrics'
                const Vertex& va = s.verts[s.idx[3 * t]], &vb = s.verts[s.idx[3 * t + 1]], &vc = s.verts[s.idx[3 * t + 2]];
                f.dudx = (dbx[0] * va.u + dbx[1] * vb.u) + dbx[2] * vc.u; f.dvdx = (dbx[0] * va.v + dbx[1] * vb.v) + dbx[2] * vc.v;
                f.dudy = (dby[0] * va.u + dby[1] * vb.u) + dby[2] * vc.u; f.dvdy = (dby[0] * va.v + dby[1] * vb.v) + dby[2] * vc.v;
            }
            float bc[4];
            base_color(c, it.mat, &f, bc);
            V3 base{bc[0], bc[1], bc[2]};
            V3 col = base;
            if (!it.mat.unlit) {
                // compute_ambient_light: base * (light_color * intensity)
                V3 amb = V3{c.ambient[0], c.ambient[1], c.ambient[2]} * c.ambient[3];
                col = base * amb;
                for (const Light& l : c.lights) {
                    V3 contrib;
                    if (light_contribution(c, l, f, base, contrib)) col = col + contrib;
                }
            }
            return col;
        };
        parallel_for(c.threads, H, [&](size_t lo, size_t hi) {
            for (size_t j = lo; j < hi; ++j)
                for (uint32_t i = 0; i < W; ++i) {
                    size_t p = j * W + i;
                    float* hdr = &c.hdr[p * 3];
                    if (S == 1) {
                        uint32_t id = (uint32_t)(c.vis[p] & 0xFFFFFFFFull);
                        if (id == RND_NONE) { hdr[0] = clear[0]; hdr[1] = clear[1]; hdr[2] = clear[2]; continue; }
                        V3 col = shade(id, i, (uint32_t)j);
                        hdr[0] = round_f16(col.x); hdr[1] = round_f16(col.y); hdr[2] = round_f16(col.z);
                        continue;
                    }
                    // every sample holds the f16 colour of its winner (or the clear colour); the resolve
                    // target gets their mean (resolve attachment, visual_server.rs:1134-1153)
                    float sum[3] = {0, 0, 0};
                    uint32_t last_id = RND_NONE; V3 last{0, 0, 0};
                    for (uint32_t smp = 0; smp < S; ++smp) {
                        uint32_t id = (uint32_t)(c.vis[(size_t)smp * P + p] & 0xFFFFFFFFull);
                        float sc[3] = {clear[0], clear[1], clear[2]};
                        if (id != RND_NONE) {
                            if (id != last_id) { last = shade(id, i, (uint32_t)j); last_id = id; }
                            sc[0] = round_f16(last.x); sc[1] = round_f16(last.y); sc[2] = round_f16(last.z);
                        }
                        for (int k = 0; k < 3; ++k) sum[k] = sum[k] + sc[k];
                    }
                    for (int k = 0; k < 3; ++k) hdr[k] = round_f16(sum[k] * 0.25f);
                }
        });
    } else {
        // forward: prepass colour of the depth winner (blend REPLACE into Rgba16Float) ...
        for (size_t p = 0; p < P; ++p) {
            float* hdr = &c.hdr[p * 3];
            uint32_t id = (uint32_t)(c.vis[p] & 0xFFFFFFFFull);
            if (id == RND_NONE) { hdr[0] = clear[0]; hdr[1] = clear[1]; hdr[2] = clear[2]; continue; }
            const DrawItem& it = items[item_of(id)];
            float bc[4];
            base_color(c, it.mat, nullptr, bc);
            V3 base{bc[0], bc[1], bc[2]};
            V3 col = base;
            if (!it.mat.unlit) col = base * (V3{c.ambient[0], c.ambient[1], c.ambient[2]} * c.ambient[3]);
            hdr[0] = round_f16(col.x); hdr[1] = round_f16(col.y); hdr[2] = round_f16(col.z);
        }
        // ... then for mesh { for light { draw } } with depth Equal, additive blend
        // (pipeline3d.rs:346-367, :507-513, :561-575)
        for (const DrawItem& it : items) {
            float bc0[4];
            base_color(c, it.mat, nullptr, bc0);
            if (bc0[3] < 0.5f || it.mat.unlit) continue;
            const Submesh& s = *it.sub;
            for (const Light& l : c.lights) {
                for (size_t t = 0; t < s.idx.size() / 3; ++t) {
                    const V4& c0 = it.clip[s.idx[3 * t]], &c1 = it.clip[s.idx[3 * t + 1]], &c2 = it.clip[s.idx[3 * t + 2]];
                    raster_triangle(c0, c1, c2, W, H, [&](uint32_t i, uint32_t j, float z) {
                        size_t p = (size_t)j * W + i;
                        if (!owns_pixel(c, i, j)) return;
                        if (fbits(z) != (uint32_t)(c.vis[p] >> 32) || !(z < 1.0f)) return;   // depth Equal
                        float b[3];
                        barycentrics(c0, c1, c2, i, j, W, H, b);
                        Frag f = frag_forward(it, c.cam, (uint32_t)t, b);
                        float bc[4];
                        base_color(c, it.mat, nullptr, bc);
                        V3 contrib;
                        if (!light_contribution(c, l, f, V3{bc[0], bc[1], bc[2]}, contrib)) return;
                        float* hdr = &c.hdr[p * 3];
                        hdr[0] = round_f16(hdr[0] + contrib.x);
                        hdr[1] = round_f16(hdr[1] + contrib.y);
                        hdr[2] = round_f16(hdr[2] + contrib.z);
                    });
                }
            }
        }
    }
    if (c.SW == W && c.SH == H) {
        parallel_for(c.threads, P, [&](size_t lo, size_t hi) {
            for (size_t p = lo; p < hi; ++p) present_pixel(c, &c.hdr[p * 3], &c.rgba[p * 4]);
        });
    } else {
        // show_texture.wgsl over the surface: uv = pixel centre / surface size; the 3D target is sampled
        // with the unfiltered sampler (nearest, ClampToEdge) at factors <= 1 and the filtered one
        // (linear, Repeat; one mip level, isotropic) above 1 (visual_server.rs:629-633), then tone-mapped
        c.rgba.assign((size_t)c.SW * c.SH * 4, 0);
        const bool linear = c.size_factor > 1.0f;
        parallel_for(c.threads, c.SH, [&](size_t lo, size_t hi) {
            for (size_t j = lo; j < hi; ++j)
                for (uint32_t i = 0; i < c.SW; ++i) {
                    const float u = ((float)i + 0.5f) / (float)c.SW, v = ((float)j + 0.5f) / (float)c.SH;
                    float px[3];
                    if (!linear) {
                        uint32_t x = std::min((uint32_t)std::floor(u * (float)W), W - 1), y = std::min((uint32_t)std::floor(v * (float)H), H - 1);
                        for (int k = 0; k < 3; ++k) px[k] = c.hdr[((size_t)y * W + x) * 3 + k];
                    } else {
                        const float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
                        const float xf = std::floor(x), yf = std::floor(y), fx = x - xf, fy = y - yf;
                        auto wrap = [](float f, uint32_t n) { long long k = (long long)f % (long long)n; return (uint32_t)(k < 0 ? k + n : k); };
                        const uint32_t x0 = wrap(xf, W), x1 = wrap(xf + 1.0f, W), y0 = wrap(yf, H), y1 = wrap(yf + 1.0f, H);
                        for (int k = 0; k < 3; ++k) {
                            const float t00 = c.hdr[((size_t)y0 * W + x0) * 3 + k], t10 = c.hdr[((size_t)y0 * W + x1) * 3 + k];
                            const float t01 = c.hdr[((size_t)y1 * W + x0) * 3 + k], t11 = c.hdr[((size_t)y1 * W + x1) * 3 + k];
                            const float top = t00 * (1.0f - fx) + t10 * fx, bot = t01 * (1.0f - fx) + t11 * fx;
                            px[k] = top * (1.0f - fy) + bot * fy;
                        }
                    }
                    present_pixel(c, px, &c.rgba[((size_t)j * c.SW + i) * 4]);
                }
        });
    }

    render_overlay(c);

    auto t_end = std::chrono::steady_clock::now();
    c.timings = rnd_timings{};
    c.timings.frame_ms = std::chrono::duration<float, std::milli>(t_end - t_begin).count();
    c.stats = rnd_frame_stats{};
    c.stats.n_vertices = nv; c.stats.n_triangles = nt; c.stats.n_pixels = P;
    c.stats.n_draw_items = (uint32_t)items.size(); c.stats.n_lights = (uint32_t)c.lights.size();
    c.have_frame = true;
    return RND_OK;
}

}  // namespace orc

// ------------------------------------------------------------------------------------
// C API: same shape as include/rnd.h with the prefix orc_ (so one harness drives both).
// ------------------------------------------------------------------------------------
using orc::Ctx;
static std::string g_create_err;
#define CTX(c) (reinterpret_cast<Ctx*>(c))
#define CCTX(c) (reinterpret_cast<const Ctx*>(c))

static orc::M4 load_m4(const float* p) { orc::M4 m; std::memcpy(m.c, p, 64); return m; }

extern "C" {

static void set_font(Ctx* c, uint32_t w, uint32_t h, const uint8_t* data) {
    orc::Texture t;
    t.w = w; t.h = h; t.mips = 1; t.opaque = false;
    t.level_offset.push_back(0);
    t.texels.resize((size_t)w * h * 4);
    for (size_t k = 0; k < (size_t)w * h * 4; ++k) t.texels[k] = (float)(data[k] / 255.0);     // Rgba8Unorm
    c->font = std::move(t);
}
int orc_create(rnd_ctx** out, uint32_t width, uint32_t height, int /*device*/) {
    if (!out || width == 0 || height == 0 || width > 16384 || height > 16384) {
        g_create_err = "orc_create: bad size";
        return RND_ERR_INVALID;
    }
    Ctx* c = new Ctx();
    c->W = c->SW = width; c->H = c->SH = height;
    c->meshes[orc::QUAD_MESH_ID] = orc::quad_mesh();
    const uint8_t yellow[4] = {255, 255, 0, 255};      // visual_server.rs:82
    set_font(c, 1, 1, yellow);
    *out = reinterpret_cast<rnd_ctx*>(c);
    return RND_OK;
}
void orc_destroy(rnd_ctx* ctx) { delete CTX(ctx); }
const char* orc_last_error(const rnd_ctx* ctx) { return ctx ? CCTX(ctx)->err.c_str() : g_create_err.c_str(); }
const char* orc_backend_name(void) { return "cpu-oracle"; }

int orc_render_size(const rnd_ctx* ctx, uint32_t* w, uint32_t* h) { *w = CCTX(ctx)->SW; *h = CCTX(ctx)->SH; return RND_OK; }
int orc_render_resolution(const rnd_ctx* ctx, uint32_t* w, uint32_t* h) { *w = CCTX(ctx)->W; *h = CCTX(ctx)->H; return RND_OK; }
static int orc_apply_size(Ctx* c, uint32_t sw, uint32_t sh, float f) {
    // recreate_render_targets (visual_server.rs:621-623): (render_size.as_vec2() * factor).as_uvec2()
    if (!(f > 0.0f) || !(f <= 16.0f)) { c->err = "render_size_factor must be in (0, 16]"; return RND_ERR_INVALID; }
    const uint32_t w = (uint32_t)((float)sw * f), h = (uint32_t)((float)sh * f);
    if (sw == 0 || sh == 0 || w == 0 || h == 0 || w > 16384 || h > 16384) { c->err = "bad size"; return RND_ERR_INVALID; }
    if (f != 1.0f && c->world > 1) { c->err = "render_size_factor != 1 on a tile partition is not built"; return RND_ERR_UNSUPPORTED; }
    c->SW = sw; c->SH = sh; c->W = w; c->H = h; c->size_factor = f; c->have_frame = false;
    return RND_OK;
}
int orc_set_render_size(rnd_ctx* ctx, uint32_t w, uint32_t h) { return orc_apply_size(CTX(ctx), w, h, CCTX(ctx)->size_factor); }
int orc_render_size_factor(const rnd_ctx* ctx, float* f) { *f = CCTX(ctx)->size_factor; return RND_OK; }
int orc_set_render_size_factor(rnd_ctx* ctx, float f) { return orc_apply_size(CTX(ctx), CCTX(ctx)->SW, CCTX(ctx)->SH, f); }
int orc_msaa_sample_count(const rnd_ctx* ctx, uint32_t* n) { *n = CCTX(ctx)->msaa; return RND_OK; }
int orc_set_msaa(rnd_ctx* ctx, uint32_t n) {
    if (n != 1 && n != 4) { CTX(ctx)->err = "sample count must be 1 or 4"; return RND_ERR_INVALID; }
    CTX(ctx)->msaa = n; CTX(ctx)->have_frame = false;
    return RND_OK;
}
int orc_tone_mapping(const rnd_ctx* ctx, uint32_t* m) { *m = CCTX(ctx)->tone; return RND_OK; }
int orc_set_tone_mapping(rnd_ctx* ctx, uint32_t m) {
    if (m > 1) { CTX(ctx)->err = "bad tone mapping"; return RND_ERR_INVALID; }
    CTX(ctx)->tone = m; return RND_OK;
}

int orc_register_mesh(rnd_ctx* ctx, uint32_t mesh_id, uint32_t n, const rnd_submesh_desc* subs) {
    Ctx* c = CTX(ctx);
    if (!subs && n) { c->err = "null submeshes"; return RND_ERR_INVALID; }
    orc::Mesh m;
    for (uint32_t k = 0; k < n; ++k) {
        const rnd_submesh_desc& d = subs[k];
        if ((d.n_vertices && !d.vertices) || (d.n_indices && !d.indices) || d.n_indices % 3) {
            c->err = "bad submesh"; return RND_ERR_INVALID;
        }
        orc::Submesh s;
        s.verts.resize(d.n_vertices);
        for (uint32_t v = 0; v < d.n_vertices; ++v) {
            const rnd_vertex& x = d.vertices[v];
            s.verts[v] = {{x.position[0], x.position[1], x.position[2]}, {x.normal[0], x.normal[1], x.normal[2]}, x.uv[0], x.uv[1]};
        }
        s.idx.assign(d.indices, d.indices + d.n_indices);
        for (uint32_t i : s.idx) if (i >= d.n_vertices) { c->err = "index out of range"; return RND_ERR_INVALID; }
        s.material = d.material_id;
        m.subs.push_back(std::move(s));
    }
    c->meshes[mesh_id] = std::move(m);
    return RND_OK;
}
int orc_register_material(rnd_ctx* ctx, uint32_t id, const float base[4], uint32_t billboard, uint32_t unlit, uint32_t tex) {
    Ctx* c = CTX(ctx);
    if (!base || billboard > 2 || id == RND_NONE) { c->err = "bad material"; return RND_ERR_INVALID; }
    if (tex != RND_NONE && !c->textures.count(tex)) { c->err = "texture id not registered"; return RND_ERR_UNKNOWN_ID; }
    c->materials[id] = orc::Material{{base[0], base[1], base[2], base[3]}, billboard, unlit ? 1u : 0u, tex};
    return RND_OK;
}
int orc_register_texture(rnd_ctx* ctx, uint32_t id, uint32_t w, uint32_t h, uint32_t mips, const uint8_t* data) {
    Ctx* c = CTX(ctx);
    if (id == RND_NONE || !w || !h || !mips || !data || w > 16384 || h > 16384) { c->err = "bad texture"; return RND_ERR_INVALID; }
    uint32_t max_levels = 1;
    while ((std::max(w, h) >> max_levels) != 0) ++max_levels;
    if (mips > max_levels) { c->err = "more mip levels than the size allows"; return RND_ERR_INVALID; }
    orc::Texture t;
    t.w = w; t.h = h; t.mips = mips; t.opaque = true;
    size_t total = 0;
    for (uint32_t l = 0; l < mips; ++l) { t.level_offset.push_back(total); total += (size_t)std::max(1u, w >> l) * std::max(1u, h >> l); }
    t.texels.resize(total * 4);
    for (size_t k = 0; k < total; ++k) {
        for (int ch = 0; ch < 3; ++ch) {                       // sRGB EOTF, evaluated in f64 and rounded once
            double v = data[4 * k + ch] / 255.0;
            double lin = v <= 0.04045 ? v / 12.92 : std::pow((v + 0.055) / 1.055, 2.4);
            t.texels[4 * k + ch] = (float)lin;
        }
        t.texels[4 * k + 3] = (float)(data[4 * k + 3] / 255.0);
        if (data[4 * k + 3] != 255) t.opaque = false;
    }
    c->textures[id] = std::move(t);
    return RND_OK;
}

int orc_set_camera(rnd_ctx* ctx, const float t[16], float vfov, float aspect, float zn, float zf) {
    Ctx* c = CTX(ctx);
    if (!t) { c->err = "null transform"; return RND_ERR_INVALID; }
    c->cam = load_m4(t);
    c->proj = orc::perspective_lh(vfov, aspect, zn, zf);
    c->view = orc::affine_inverse(c->cam);
    c->inv_pv = orc::inverse(orc::mul(c->proj, c->view));        // visual_server.rs:210
    return RND_OK;
}
int orc_set_light(rnd_ctx* ctx, uint64_t id, const float t[16], const float color[4], uint32_t kind, float radius) {
    Ctx* c = CTX(ctx);
    if (!t || !color || kind > 1) { c->err = "bad light"; return RND_ERR_INVALID; }
    orc::Light l{id, load_m4(t), {color[0], color[1], color[2], color[3]}, kind, kind == RND_LIGHT_POINT ? radius : 0.0f, {}, {}};
    // cascades follow the camera as of this call (create_light / update_light, visual_server.rs:380-381, :445-446)
    orc::shadow_cascade_projviews(c->inv_pv, orc::V3{l.transform.c[8], l.transform.c[9], l.transform.c[10]}, l.cascades);
    for (auto& e : c->lights) if (e.id == id) { e = l; return RND_OK; }
    c->lights.push_back(l);
    return RND_OK;
}
int orc_set_mesh_instance_ex(rnd_ctx* ctx, uint64_t id, const float t[16], uint32_t mesh, uint32_t mat, uint32_t shadows) {
    Ctx* c = CTX(ctx);
    if (!t) { c->err = "null transform"; return RND_ERR_INVALID; }
    if (!c->meshes.count(mesh)) { c->err = "mesh id not registered"; return RND_ERR_UNKNOWN_ID; }
    if (mat != RND_NONE && !c->materials.count(mat)) { c->err = "material id not registered"; return RND_ERR_UNKNOWN_ID; }
    orc::Instance inst{id, load_m4(t), mesh, mat, shadows != 0};
    for (auto& e : c->instances) if (e.id == id) { e = inst; return RND_OK; }
    c->instances.push_back(inst);
    return RND_OK;
}
int orc_set_mesh_instance(rnd_ctx* ctx, uint64_t id, const float t[16], uint32_t mesh) {
    return orc_set_mesh_instance_ex(ctx, id, t, mesh, RND_NONE, 1);
}
int orc_set_mesh_instances(rnd_ctx* ctx, uint32_t n, const uint64_t* ids, const float* t, const uint32_t* meshes) {
    for (uint32_t k = 0; k < n; ++k) {
        int rc = orc_set_mesh_instance_ex(ctx, ids[k], t + 16 * (size_t)k, meshes[k], RND_NONE, 1);
        if (rc != RND_OK) return rc;
    }
    return RND_OK;
}
int orc_set_sprite(rnd_ctx* ctx, uint64_t id, const float t[16], uint32_t tex, const float color[4]) {
    Ctx* c = CTX(ctx);
    if (!t || !color) { c->err = "bad sprite"; return RND_ERR_INVALID; }
    if (tex != RND_NONE && !c->textures.count(tex)) { c->err = "texture id not registered"; return RND_ERR_UNKNOWN_ID; }
    uint32_t mid;
    auto it = c->sprite_material.find(id);
    if (it == c->sprite_material.end()) { mid = c->next_sprite_material++; c->sprite_material[id] = mid; }
    else mid = it->second;
    c->materials[mid] = orc::Material{{color[0], color[1], color[2], color[3]}, RND_BILLBOARD_ON, 1u, tex};
    return orc_set_mesh_instance_ex(ctx, id, t, orc::QUAD_MESH_ID, mid, 0);
}
int orc_reset_scene(rnd_ctx* ctx) {
    Ctx* c = CTX(ctx);
    c->meshes.clear(); c->materials.clear(); c->textures.clear(); c->lights.clear(); c->instances.clear();
    c->sprite_material.clear();
    c->meshes[orc::QUAD_MESH_ID] = orc::quad_mesh();
    c->fullscreen = 0;          // RenderScene::fullscreen_texture (visual_server.rs:903)
    c->inv_pv = orc::identity();    // RenderScene::default() (visual_server.rs:579-581, :896-898): cascades of a light set before the next set_camera come from the identity
    return RND_OK;
}

int orc_set_uiboxes(rnd_ctx* ctx, uint32_t n, const rnd_uibox* boxes) {
    Ctx* c = CTX(ctx);
    if (n && !boxes) { c->err = "null boxes"; return RND_ERR_INVALID; }
    c->uiboxes.assign(boxes, boxes + n);
    return RND_OK;
}
int orc_set_font_image(rnd_ctx* ctx, uint32_t w, uint32_t h, const uint8_t* data) {
    Ctx* c = CTX(ctx);
    if (!w || !h || !data || w > 16384 || h > 16384) { c->err = "bad font image"; return RND_ERR_INVALID; }
    set_font(c, w, h, data);
    return RND_OK;
}
int orc_reset_texts(rnd_ctx* ctx) { CTX(ctx)->glyphs.clear(); return RND_OK; }
int orc_add_text(rnd_ctx* ctx, const uint8_t* text, uint32_t n, float x, float y, float font_size) {
    Ctx* c = CTX(ctx);
    if (n && !text) { c->err = "null text"; return RND_ERR_INVALID; }
    for (uint32_t i = 0; i < n; ++i) {          // visual_server.rs:556-570
        Ctx::Glyph g;
        g.pos[0] = x + (float)i * font_size * 1.1667f * 0.5f; g.pos[1] = y + 0.0f;
        g.scale[0] = font_size * 1.1667f * 0.5f; g.scale[1] = font_size;
        g.glyph = std::min<uint32_t>(text[i], 127u);
        c->glyphs.push_back(g);
    }
    return RND_OK;
}
int orc_set_depth_fullscreen_texture(rnd_ctx* ctx) { CTX(ctx)->fullscreen = 1; return RND_OK; }
int orc_set_shadow_map_fullscreen_texture(rnd_ctx* ctx, uint64_t light_id) {
    Ctx* c = CTX(ctx);
    for (const auto& l : c->lights)
        if (l.id == light_id) { c->fullscreen = 2; c->fullscreen_light = light_id; return RND_OK; }
    return RND_OK;      // "warning: no such light registered" and return (visual_server.rs:329-332)
}
int orc_unset_fullscreen_texture(rnd_ctx* ctx) { CTX(ctx)->fullscreen = 0; return RND_OK; }

int orc_render(rnd_ctx* ctx) { return orc::render(*CTX(ctx)); }
int orc_finish(rnd_ctx*) { return RND_OK; }

#define NEED_FRAME(c) if (!(c)->have_frame) { (c)->err = "no frame rendered yet"; return RND_ERR_NO_FRAME; }
int orc_read_color_rgba8(rnd_ctx* ctx, uint8_t* dst) { Ctx* c = CTX(ctx); NEED_FRAME(c); std::memcpy(dst, c->rgba.data(), c->rgba.size()); return RND_OK; }
int orc_read_overlay_rgba8(rnd_ctx* ctx, uint8_t* dst) { Ctx* c = CTX(ctx); NEED_FRAME(c); std::memcpy(dst, c->overlay.data(), c->overlay.size()); return RND_OK; }
int orc_read_color_hdr(rnd_ctx* ctx, float* dst) { Ctx* c = CTX(ctx); NEED_FRAME(c); std::memcpy(dst, c->hdr.data(), c->hdr.size() * 4); return RND_OK; }
int orc_read_depth(rnd_ctx* ctx, float* dst) {
    Ctx* c = CTX(ctx); NEED_FRAME(c);
    for (size_t p = 0; p < (size_t)c->W * c->H; ++p) dst[p] = orc::bitsf((uint32_t)(c->vis[p] >> 32));
    return RND_OK;
}
int orc_read_coverage(rnd_ctx* ctx, uint8_t* dst) {
    Ctx* c = CTX(ctx); NEED_FRAME(c);
    for (size_t p = 0; p < (size_t)c->W * c->H; ++p) dst[p] = (uint32_t)(c->vis[p] & 0xFFFFFFFFull) != RND_NONE;
    return RND_OK;
}
int orc_read_triangle_ids(rnd_ctx* ctx, uint32_t* dst) {
    Ctx* c = CTX(ctx); NEED_FRAME(c);
    for (size_t p = 0; p < (size_t)c->W * c->H; ++p) dst[p] = (uint32_t)(c->vis[p] & 0xFFFFFFFFull);
    return RND_OK;
}
int orc_last_timings(rnd_ctx* ctx, rnd_timings* out) { *out = CTX(ctx)->timings; return RND_OK; }
int orc_frame_stats_get(rnd_ctx* ctx, rnd_frame_stats* out) { *out = CTX(ctx)->stats; return RND_OK; }

int orc_set_tile_partition(rnd_ctx* ctx, uint32_t rank, uint32_t world) {
    Ctx* c = CTX(ctx);
    if (world == 0 || rank >= world) { c->err = "bad partition"; return RND_ERR_INVALID; }
    if (world > 1 && c->size_factor != 1.0f) { c->err = "render_size_factor != 1 on a tile partition is not built"; return RND_ERR_UNSUPPORTED; }
    c->rank = rank; c->world = world; return RND_OK;
}
int orc_tile_info(const rnd_ctx* ctx, uint32_t* tile_px, uint32_t* n_owned, uint32_t* n_total) {
    const Ctx* c = CCTX(ctx);
    uint32_t tx = (c->W + orc::TILE - 1) / orc::TILE, ty = (c->H + orc::TILE - 1) / orc::TILE, own = 0;
    for (uint32_t y = 0; y < ty; ++y) for (uint32_t x = 0; x < tx; ++x) own += orc::owns_tile(*c, x, y);
    *tile_px = orc::TILE; *n_owned = own; *n_total = tx * ty; return RND_OK;
}
/* oracle-only knobs */
int orc_set_mode(rnd_ctx* ctx, int mode) { CTX(ctx)->mode = mode ? 1 : 0; return RND_OK; }
int orc_set_threads(rnd_ctx* ctx, int n) { CTX(ctx)->threads = n < 1 ? 1 : n; return RND_OK; }
/* host-side matrices, for checking the product's host logic */
int orc_get_matrices(rnd_ctx* ctx, float proj[16], float view[16], float projview[16]) {
    Ctx* c = CTX(ctx);
    std::memcpy(proj, c->proj.c, 64); std::memcpy(view, c->view.c, 64);
    orc::M4 pv = orc::mul(c->proj, c->view); std::memcpy(projview, pv.c, 64);
    return RND_OK;
}
/* the four world -> light-clip matrices of a light's shadow cascades, as of its last set_light */
int orc_get_light_cascades(rnd_ctx* ctx, uint64_t id, float out[64]) {
    Ctx* c = CTX(ctx);
    for (const auto& l : c->lights)
        if (l.id == id) { for (int k = 0; k < 4; ++k) std::memcpy(out + 16 * k, l.cascades[k].c, 64); return RND_OK; }
    c->err = "no such light";
    return RND_ERR_UNKNOWN_ID;
}
/* tile-major RGBA8 of the owned tiles, as the product's rnd_tiles_device_ptr buffer (host copy) */
int orc_read_tiles(rnd_ctx* ctx, uint8_t* dst) {
    Ctx* c = CTX(ctx); NEED_FRAME(c);
    uint32_t tx = (c->W + orc::TILE - 1) / orc::TILE, ty = (c->H + orc::TILE - 1) / orc::TILE;
    size_t k = 0;
    for (uint32_t y = 0; y < ty; ++y) for (uint32_t x = 0; x < tx; ++x) {
        if (!orc::owns_tile(*c, x, y)) continue;
        for (uint32_t j = 0; j < orc::TILE; ++j) for (uint32_t i = 0; i < orc::TILE; ++i) {
            uint32_t px = x * orc::TILE + i, py = y * orc::TILE + j;
            for (int ch = 0; ch < 4; ++ch)
                dst[k++] = (px < c->W && py < c->H) ? c->rgba[((size_t)py * c->W + px) * 4 + ch] : 0;
        }
    }
    return RND_OK;
}

}  // extern "C"
