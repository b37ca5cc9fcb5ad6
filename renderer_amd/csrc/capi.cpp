// extern "C" boundary of include/rnd.h over rnd::VisualServer.  Exceptions never cross it: every
// entry point returns an rnd_status and keeps the message for rnd_last_error().
#include <cstring>
#include <string>

#include "../../include/rnd.h"
#include "visual_server.hpp"

struct rnd_ctx {
    rnd::VisualServer* vs = nullptr;
    std::string err;
};

static thread_local std::string g_create_error;

template <class F>
static int guard(rnd_ctx* ctx, F&& f) {
    if (!ctx || !ctx->vs) return RND_ERR_INVALID;
    try {
        f(*ctx->vs);
        return RND_OK;
    } catch (const rnd::Error& e) {
        ctx->err = e.what();
        return e.status;
    } catch (const std::bad_alloc&) {
        ctx->err = "host allocation failed";
        return RND_ERR_OOM;
    } catch (const std::exception& e) {
        ctx->err = e.what();
        return RND_ERR_DEVICE;
    }
}
template <class F>
static int guard(const rnd_ctx* ctx, F&& f) { return guard(const_cast<rnd_ctx*>(ctx), f); }

#define REQUIRE(ctx, cond, msg) \
    if (!(cond)) { if (ctx) (ctx)->err = (msg); return RND_ERR_INVALID; }

extern "C" {

int rnd_create(rnd_ctx** out, uint32_t width, uint32_t height, int device) {
    if (!out) { g_create_error = "rnd_create: null output pointer"; return RND_ERR_INVALID; }
    *out = nullptr;
    try {
        rnd_ctx* c = new rnd_ctx();
        try {
            c->vs = new rnd::VisualServer(width, height, device);
        } catch (...) {
            delete c;
            throw;
        }
        *out = c;
        return RND_OK;
    } catch (const rnd::Error& e) {
        g_create_error = e.what();
        return e.status;
    } catch (const std::exception& e) {
        g_create_error = e.what();
        return RND_ERR_DEVICE;
    }
}

void rnd_destroy(rnd_ctx* ctx) {
    if (!ctx) return;
    delete ctx->vs;
    delete ctx;
}

const char* rnd_last_error(const rnd_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }
const char* rnd_backend_name(void) { return "hip-gfx950"; }

int rnd_render_size(const rnd_ctx* ctx, uint32_t* w, uint32_t* h) {
    REQUIRE(const_cast<rnd_ctx*>(ctx), w && h, "null output");
    return guard(ctx, [&](rnd::VisualServer& v) { v.render_size(*w, *h); });
}
int rnd_render_resolution(const rnd_ctx* ctx, uint32_t* w, uint32_t* h) {
    REQUIRE(const_cast<rnd_ctx*>(ctx), w && h, "null output");
    return guard(ctx, [&](rnd::VisualServer& v) { v.render_resolution(*w, *h); });
}
int rnd_set_render_size(rnd_ctx* ctx, uint32_t w, uint32_t h) { return guard(ctx, [&](rnd::VisualServer& v) { v.set_render_size(w, h); }); }
int rnd_render_size_factor(const rnd_ctx* ctx, float* f) {
    REQUIRE(const_cast<rnd_ctx*>(ctx), f, "null output");
    return guard(ctx, [&](rnd::VisualServer& v) { *f = v.render_size_factor(); });
}
int rnd_set_render_size_factor(rnd_ctx* ctx, float f) { return guard(ctx, [&](rnd::VisualServer& v) { v.set_render_size_factor(f); }); }
int rnd_msaa_sample_count(const rnd_ctx* ctx, uint32_t* n) {
    REQUIRE(const_cast<rnd_ctx*>(ctx), n, "null output");
    return guard(ctx, [&](rnd::VisualServer& v) { *n = v.msaa_sample_count(); });
}
int rnd_set_msaa(rnd_ctx* ctx, uint32_t n) { return guard(ctx, [&](rnd::VisualServer& v) { v.set_msaa(n); }); }
int rnd_tone_mapping(const rnd_ctx* ctx, uint32_t* m) {
    REQUIRE(const_cast<rnd_ctx*>(ctx), m, "null output");
    return guard(ctx, [&](rnd::VisualServer& v) { *m = v.tone_mapping(); });
}
int rnd_set_tone_mapping(rnd_ctx* ctx, uint32_t m) { return guard(ctx, [&](rnd::VisualServer& v) { v.set_tone_mapping(m); }); }

int rnd_register_mesh(rnd_ctx* ctx, uint32_t mesh_id, uint32_t n, const rnd_submesh_desc* subs) {
    return guard(ctx, [&](rnd::VisualServer& v) { v.register_mesh(mesh_id, n, subs); });
}
int rnd_register_material(rnd_ctx* ctx, uint32_t id, const float base_color[4], uint32_t billboard_mode, uint32_t unlit,
                          uint32_t texture_id) {
    REQUIRE(ctx, base_color, "null base colour");
    return guard(ctx, [&](rnd::VisualServer& v) {
        rnd::Material m;
        std::memcpy(m.base_color, base_color, 16);
        m.billboard_mode = billboard_mode; m.unlit = unlit != 0; m.texture = texture_id;
        v.register_material(id, m);
    });
}
int rnd_register_texture(rnd_ctx* ctx, uint32_t id, uint32_t w, uint32_t h, uint32_t mips, const uint8_t* data) {
    return guard(ctx, [&](rnd::VisualServer& v) { v.register_texture(id, w, h, mips, data); });
}

int rnd_set_camera(rnd_ctx* ctx, const float t[16], float vfov, float aspect, float z_near, float z_far) {
    REQUIRE(ctx, t, "null transform");
    return guard(ctx, [&](rnd::VisualServer& v) {
        v.set_camera(rnd::Mat4::from_cols_array(t), rnd::Camera{vfov, aspect, z_near, z_far});
    });
}
int rnd_set_light(rnd_ctx* ctx, uint64_t id, const float t[16], const float color[4], uint32_t kind, float radius) {
    REQUIRE(ctx, t && color, "null transform or colour");
    return guard(ctx, [&](rnd::VisualServer& v) {
        rnd::Light l;
        std::memcpy(l.color, color, 16);
        l.kind = kind; l.radius = radius;
        v.set_light(id, rnd::Mat4::from_cols_array(t), l);
    });
}
int rnd_set_mesh_instance(rnd_ctx* ctx, uint64_t id, const float t[16], uint32_t mesh_id) {
    REQUIRE(ctx, t, "null transform");
    return guard(ctx, [&](rnd::VisualServer& v) { v.set_mesh_instance(id, rnd::Mat4::from_cols_array(t), mesh_id, RND_NONE, true); });
}
int rnd_set_mesh_instance_ex(rnd_ctx* ctx, uint64_t id, const float t[16], uint32_t mesh_id, uint32_t material_override,
                             uint32_t casts_shadows) {
    REQUIRE(ctx, t, "null transform");
    return guard(ctx, [&](rnd::VisualServer& v) {
        v.set_mesh_instance(id, rnd::Mat4::from_cols_array(t), mesh_id, material_override, casts_shadows != 0);
    });
}
int rnd_set_sprite(rnd_ctx* ctx, uint64_t id, const float t[16], uint32_t texture_id, const float color[4]) {
    REQUIRE(ctx, t && color, "null transform or colour");
    return guard(ctx, [&](rnd::VisualServer& v) { v.set_sprite(id, rnd::Mat4::from_cols_array(t), texture_id, color); });
}
int rnd_set_mesh_instances(rnd_ctx* ctx, uint32_t n, const uint64_t* ids, const float* transforms_cols, const uint32_t* mesh_ids) {
    REQUIRE(ctx, !n || (ids && transforms_cols && mesh_ids), "null instance arrays");
    return guard(ctx, [&](rnd::VisualServer& v) {
        for (uint32_t k = 0; k < n; ++k) v.set_mesh_instance(ids[k], rnd::Mat4::from_cols_array(transforms_cols + 16 * (size_t)k), mesh_ids[k], RND_NONE, true);
    });
}
int rnd_reset_scene(rnd_ctx* ctx) { return guard(ctx, [&](rnd::VisualServer& v) { v.reset_scene(); }); }

int rnd_set_uiboxes(rnd_ctx* ctx, uint32_t n, const rnd_uibox* boxes) { return guard(ctx, [&](rnd::VisualServer& v) { v.set_uiboxes(n, boxes); }); }
int rnd_set_font_image(rnd_ctx* ctx, uint32_t w, uint32_t h, const uint8_t* rgba8) { return guard(ctx, [&](rnd::VisualServer& v) { v.set_font_image(w, h, rgba8); }); }
int rnd_reset_texts(rnd_ctx* ctx) { return guard(ctx, [&](rnd::VisualServer& v) { v.reset_texts(); }); }
int rnd_add_text(rnd_ctx* ctx, const uint8_t* text, uint32_t n, float x, float y, float font_size) {
    return guard(ctx, [&](rnd::VisualServer& v) { v.add_text(text, n, x, y, font_size); });
}
int rnd_set_depth_fullscreen_texture(rnd_ctx* ctx) { return guard(ctx, [&](rnd::VisualServer& v) { v.set_depth_fullscreen_texture(); }); }
int rnd_set_shadow_map_fullscreen_texture(rnd_ctx* ctx, uint64_t light_id) { return guard(ctx, [&](rnd::VisualServer& v) { v.set_shadow_map_fullscreen_texture(light_id); }); }
int rnd_unset_fullscreen_texture(rnd_ctx* ctx) { return guard(ctx, [&](rnd::VisualServer& v) { v.unset_fullscreen_texture(); }); }
int rnd_read_overlay_rgba8(rnd_ctx* ctx, uint8_t* dst) {
    REQUIRE(ctx, dst, "null destination");
    return guard(ctx, [&](rnd::VisualServer& v) { v.read_overlay_rgba8(dst); });
}

int rnd_render(rnd_ctx* ctx) { return guard(ctx, [&](rnd::VisualServer& v) { v.render(); }); }
int rnd_render_noalloc(rnd_ctx* ctx, void* out_dev, uint64_t out_bytes) {
    return guard(ctx, [&](rnd::VisualServer& v) { v.render_noalloc(out_dev, out_bytes); });
}
int rnd_finish(rnd_ctx* ctx) { return guard(ctx, [&](rnd::VisualServer& v) { v.finish(); }); }

int rnd_read_color_rgba8(rnd_ctx* ctx, uint8_t* dst) { return guard(ctx, [&](rnd::VisualServer& v) { v.read_color_rgba8(dst); }); }
int rnd_read_color_hdr(rnd_ctx* ctx, float* dst) { return guard(ctx, [&](rnd::VisualServer& v) { v.read_color_hdr(dst); }); }
int rnd_read_depth(rnd_ctx* ctx, float* dst) {
    REQUIRE(ctx, dst, "null destination");
    return guard(ctx, [&](rnd::VisualServer& v) { v.read_planes(dst, nullptr, nullptr); });
}
int rnd_read_coverage(rnd_ctx* ctx, uint8_t* dst) {
    REQUIRE(ctx, dst, "null destination");
    return guard(ctx, [&](rnd::VisualServer& v) { v.read_planes(nullptr, nullptr, dst); });
}
int rnd_read_triangle_ids(rnd_ctx* ctx, uint32_t* dst) {
    REQUIRE(ctx, dst, "null destination");
    return guard(ctx, [&](rnd::VisualServer& v) { v.read_planes(nullptr, dst, nullptr); });
}
int rnd_last_timings(rnd_ctx* ctx, rnd_timings* out) {
    REQUIRE(ctx, out, "null output");
    return guard(ctx, [&](rnd::VisualServer& v) { *out = v.last_timings(); });
}
int rnd_mean_timings(rnd_ctx* ctx, rnd_timings* mean, uint32_t* n_frames, int reset) {
    REQUIRE(ctx, mean && n_frames, "null output");
    return guard(ctx, [&](rnd::VisualServer& v) { *mean = v.mean_timings(*n_frames, reset != 0); });
}
int rnd_time_frames(rnd_ctx* ctx, uint32_t n) { return guard(ctx, [&](rnd::VisualServer& v) { v.time_frames(n); }); }
int rnd_timing_stats(rnd_ctx* ctx, rnd_timings* mean, rnd_timings* max, uint32_t* n_frames, int reset) {
    REQUIRE(ctx, n_frames, "null output");
    return guard(ctx, [&](rnd::VisualServer& v) {
        rnd_timings mx{};
        const rnd_timings m = v.mean_timings(*n_frames, reset != 0, &mx);
        if (mean) *mean = m;
        if (max) *max = mx;
    });
}
int rnd_frame_stats_get(rnd_ctx* ctx, rnd_frame_stats* out) {
    REQUIRE(ctx, out, "null output");
    return guard(ctx, [&](rnd::VisualServer& v) { *out = v.frame_stats(); });
}

int rnd_set_tile_partition(rnd_ctx* ctx, uint32_t rank, uint32_t world) {
    return guard(ctx, [&](rnd::VisualServer& v) { v.set_tile_partition(rank, world); });
}
int rnd_tile_info(const rnd_ctx* ctx, uint32_t* tile_px, uint32_t* n_owned, uint32_t* n_total) {
    REQUIRE(const_cast<rnd_ctx*>(ctx), tile_px && n_owned && n_total, "null output");
    return guard(ctx, [&](rnd::VisualServer& v) { v.tile_info(*tile_px, *n_owned, *n_total); });
}
int rnd_tiles_device_ptr(rnd_ctx* ctx, void** p, uint64_t* n) {
    REQUIRE(ctx, p && n, "null output");
    return guard(ctx, [&](rnd::VisualServer& v) { v.tiles_device_ptr(p, n); });
}
int rnd_set_tiles_output(rnd_ctx* ctx, void* p, uint64_t n) { return guard(ctx, [&](rnd::VisualServer& v) { v.set_tiles_output(p, n); }); }
int rnd_detile(rnd_ctx* ctx, uint32_t src_rank, uint32_t world, const void* dev_tiles) {
    return guard(ctx, [&](rnd::VisualServer& v) { v.detile(src_rank, world, dev_tiles); });
}
int rnd_detile_gathered(rnd_ctx* ctx, uint32_t world, const void* dev_gathered, uint64_t stride_bytes) {
    return guard(ctx, [&](rnd::VisualServer& v) { v.detile_gathered(world, dev_gathered, stride_bytes); });
}
int rnd_packed_device_ptr(rnd_ctx* ctx, void** p, uint64_t* n) {
    REQUIRE(ctx, p && n, "null output");
    return guard(ctx, [&](rnd::VisualServer& v) { v.packed_device_ptr(p, n); });
}
int rnd_unpack(rnd_ctx* ctx, const void* dev_packed) { return guard(ctx, [&](rnd::VisualServer& v) { v.unpack(dev_packed); }); }
int rnd_set_packed_output(rnd_ctx* ctx, void* p, uint64_t n) { return guard(ctx, [&](rnd::VisualServer& v) { v.set_packed_output(p, n); }); }
int rnd_min_packed(rnd_ctx* ctx, const void* strips, uint32_t n, uint64_t stride_words, void* out, uint64_t count) {
    return guard(ctx, [&](rnd::VisualServer& v) { v.min_packed(strips, n, stride_words, out, count); });
}
int rnd_comm_make_id(uint8_t id[RND_COMM_ID_BYTES]) {
    if (!id) { g_create_error = "rnd_comm_make_id: null id"; return RND_ERR_INVALID; }
    try {
        rnd::comm_make_id(id);
        return RND_OK;
    } catch (const rnd::Error& e) {
        g_create_error = e.what();
        return e.status;
    } catch (const std::exception& e) {
        g_create_error = e.what();
        return RND_ERR_DEVICE;
    }
}
int rnd_comm_init(rnd_ctx* ctx, const uint8_t id[RND_COMM_ID_BYTES], uint32_t rank, uint32_t world) {
    return guard(ctx, [&](rnd::VisualServer& v) { v.comm_init(id, rank, world); });
}
int rnd_comm_destroy(rnd_ctx* ctx) { return guard(ctx, [&](rnd::VisualServer& v) { v.comm_destroy(); }); }
int rnd_gather_tiles(rnd_ctx* ctx) { return guard(ctx, [&](rnd::VisualServer& v) { v.gather_tiles(); }); }
int rnd_composite_packed(rnd_ctx* ctx) { return guard(ctx, [&](rnd::VisualServer& v) { v.composite_packed(); }); }
int rnd_set_stream(rnd_ctx* ctx, void* s) { return guard(ctx, [&](rnd::VisualServer& v) { v.set_stream(static_cast<hipStream_t>(s)); }); }
int rnd_color_device_ptr(rnd_ctx* ctx, void** p, uint64_t* n) {
    REQUIRE(ctx, p && n, "null output");
    return guard(ctx, [&](rnd::VisualServer& v) { v.color_device_ptr(p, n); });
}

}  // extern "C"
