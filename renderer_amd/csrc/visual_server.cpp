// Host frame driver: see visual_server.hpp.  reference: src/renderer/visual_server.rs,
// src/renderer/pipeline3d.rs:261-368 (pass structure), src/renderer/backend.rs:455-498 (present).
#include "visual_server.hpp"

#include <algorithm>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <limits>
#include <mutex>

namespace rnd {

void check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return;
    std::string msg = std::string(what) + ": " + hipGetErrorString(e);
    (void)hipGetLastError();
    throw Error(e == hipErrorOutOfMemory ? RND_ERR_OOM : RND_ERR_DEVICE, msg);
}
#define HIP_CHECK(x) check_hip((x), #x)

template <class T>
void DeviceBuffer<T>::ensure(size_t n, bool keep) {
    if (n <= count) return;
    T* p = nullptr;
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T)));
    if (ptr) {
        if (keep && count) HIP_CHECK(hipMemcpy(p, ptr, count * sizeof(T), hipMemcpyDeviceToDevice));
        (void)hipFree(ptr);   // waits for work that may still use the old allocation
    }
    ptr = p;
    count = n;
}
template <class T>
void DeviceBuffer<T>::release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    count = 0;
}

static constexpr uint32_t kQuadMeshId = 0xFFFFFFFEu;

// ---- the scratch array of the no-allocation frame (rnd_render_noalloc) --------------------------
// Every per-frame intermediate stream of a frame rendered through rnd_render_noalloc (table block, clip
// positions, snapped vertices, visibility words, setup records, bin entries, queues, counters) is carved
// out of this one file-scope array instead of hipMalloc'ed buffers.  Each GPU the library is used on holds
// its own copy at its own address until the process ends.  In host code the array's NAME is only a stand-in
// for hipGetSymbolAddress: the per-device address is looked up once per device (below, from the constructor,
// with that device current) and every launch, memset and copy takes that pointer.
static __device__ __align__(256) unsigned char scratch[VisualServer::kScratchBytes];
static constexpr int kMaxScratchDevices = 64;
static void* g_scratch_ptr[kMaxScratchDevices] = {};
// Who holds a device's array: the context of the last no-allocation frame and an event recorded behind that frame's
// last kernel.  Another context that asks for the array while the event is pending is refused (RND_ERR_INVALID) instead
// of racing; once it takes the array, the previous holder's frame is gone (its read-backs say RND_ERR_NO_FRAME).
static hipEvent_t g_scratch_busy[kMaxScratchDevices] = {};
static VisualServer* g_scratch_owner[kMaxScratchDevices] = {};
static std::mutex g_scratch_mutex;

// call with `device` current
static uint8_t* scratch_of_device(int device) {
    if (device < 0 || device >= kMaxScratchDevices) throw Error(RND_ERR_INVALID, "device ordinal beyond the scratch table");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* ptr = nullptr;
        HIP_CHECK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) throw Error(RND_ERR_DEVICE, "hipGetSymbolAddress returned a null scratch address");
        HIP_CHECK(hipEventCreateWithFlags(&g_scratch_busy[device], hipEventDisableTiming));
        g_scratch_ptr[device] = ptr;
    }
    return static_cast<uint8_t*>(g_scratch_ptr[device]);
}

VisualServer::VisualServer(uint32_t width, uint32_t height, int device) : device_(device), width_(width), height_(height) {
    if (width == 0 || height == 0 || width > kMaxRenderSize || height > kMaxRenderSize)
        throw Error(RND_ERR_INVALID, "render size must be in 1..8192");
    int n_dev = 0;
    HIP_CHECK(hipGetDeviceCount(&n_dev));
    if (device < 0 || device >= n_dev) throw Error(RND_ERR_INVALID, "no such HIP device");
    surface_w_ = width; surface_h_ = height;
    use_device();
    HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    for (auto& es : event_sets_)
        for (auto& e : es.ev) HIP_CHECK(hipEventCreate(&e));
    for (auto& s : staging_) HIP_CHECK(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    scratch_ = scratch_of_device(device_);
    // pinned blocks sized so that the no-allocation frame never has to grow them
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h_counters_), kPinnedCounterSets * sizeof(FrameCounters), hipHostMallocDefault));
    h_counter_sets_ = kPinnedCounterSets;
    for (auto& st : staging_) {
        HIP_CHECK(hipHostMalloc(&st.host, kStagingBytes, hipHostMallocMapped | hipHostMallocPortable));
        HIP_CHECK(hipHostGetDevicePointer(&st.host_dev, st.host, 0));
        st.bytes = kStagingBytes;
    }
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h_reports_), kReportSlots * sizeof(FrameReport), hipHostMallocMapped | hipHostMallocPortable));
    std::memset(h_reports_, 0, kReportSlots * sizeof(FrameReport));
    HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&d_reports_), h_reports_, 0));
    if (const char* e = std::getenv("RND_TEST_SMALL_STREAMS")) test_small_streams_ = std::atoi(e) != 0;
    d_counters_[0].ensure(1); d_counters_[1].ensure(1);
    HIP_CHECK(hipStreamCreateWithFlags(&geo_stream_, hipStreamNonBlocking));     // (a higher or lower stream priority changes nothing measurable)
    for (int k = 0; k < 2; ++k) {
        HIP_CHECK(hipEventCreateWithFlags(&geo_done_[k], hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&shade_done_[k], hipEventDisableTiming));
        if (k == 0) HIP_CHECK(hipEventCreateWithFlags(&switch_ev_, hipEventDisableTiming));
    }
    if (const char* e = std::getenv("RND_OVERLAP")) overlap_ = std::atoi(e) != 0;
    if (const char* e = std::getenv("RND_TUNE")) tune_enabled_ = std::atoi(e) != 0;
    if (const char* e = std::getenv("RND_SHADOW_STREAM")) shadow_concurrent_ = std::atoi(e) != 0;
    HIP_CHECK(hipStreamCreateWithFlags(&shadow_stream_, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreateWithFlags(&tables_ready_, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&shadows_done_, hipEventDisableTiming));
    for (hipEvent_t& e : tune_ev_) HIP_CHECK(hipEventCreate(&e));
    if (const char* e = std::getenv("RND_NO_CHAIN")) allow_no_chain_ = std::atoi(e) != 0;
    // SceneUniform defaults — visual_server.rs:63-68
    projection_ = perspective_lh(1.3f, 1.667f, 0.05f, 100.0f);
    view_ = Mat4::identity();
    camera_transform_ = Mat4::identity();
    inv_projection_view_ = Mat4::identity();     // RenderScene::default() (visual_server.rs:896-898): identity until set_camera
    rebuild_tiles();
    // initialize_default_resources — visual_server.rs:865-873 (unit quad; default material is implicit)
    static const rnd_vertex quad[4] = {{{-0.5f, 0.5f, 0.0f}, {0, 0, -1}, {0, 0}}, {{0.5f, 0.5f, 0.0f}, {0, 0, -1}, {1, 0}},
                                       {{-0.5f, -0.5f, 0.0f}, {0, 0, -1}, {0, 1}}, {{0.5f, -0.5f, 0.0f}, {0, 0, -1}, {1, 1}}};
    static const uint32_t quad_idx[6] = {0, 2, 1, 1, 2, 3};
    rnd_submesh_desc d{quad, 4, quad_idx, 6, RND_NONE};
    register_mesh(kQuadMeshId, 1, &d);
    // font texture default (visual_server.rs:82) and the sRGB decode table of the 2D blends
    static const uint8_t yellow[4] = {255, 255, 0, 255};
    set_font_image(1, 1, yellow);
    // [0, 256): sRGB EOTF in f64, rounded once (the 2D blends and the texel decode); [256, 512): alpha / 255
    float lut[512];
    for (int v = 0; v < 256; ++v) {
        const double c = v / 255.0;
        lut[v] = (float)(c <= 0.04045 ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4));
        lut[256 + v] = (float)(v / 255.0);
    }
    d_srgb_lut_.ensure(512);
    HIP_CHECK(hipMemcpy(d_srgb_lut_.ptr, lut, sizeof lut, hipMemcpyHostToDevice));
}

VisualServer::~VisualServer() {
    (void)hipSetDevice(device_);
    {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (device_ >= 0 && device_ < kMaxScratchDevices && g_scratch_owner[device_] == this) g_scratch_owner[device_] = nullptr;
    }
    try { comm_destroy(); } catch (...) {}
    if (geo_stream_) (void)hipStreamSynchronize(geo_stream_);
    if (shadow_stream_) (void)hipStreamSynchronize(shadow_stream_);
    if (stream_) (void)hipStreamSynchronize(stream_);
    shade_profile_dump();
    shadow_geo_.release();
    for (auto& kv : meshes_)
        for (auto& s : kv.second.submeshes) { s.vertex_buffer.release(); s.positions.release(); s.index_buffer.release(); s.chunks.release(); }
    for (int k = 0; k < 2; ++k) { d_tables_[k].release(); d_clip_[k].release(); d_vis_[k].release(); d_counters_[k].release(); d_stat_stripes_[k].release(); }
    d_snap_.release(); d_hiz_.release(); d_unit_bounds_.release(); d_far_units_.release(); d_wide_list_.release(); d_tile_slices_.release(); d_raster_extra_.release();
    d_records_.release(); d_tilewords_.release(); d_clip_queue_.release(); d_chunk_count_.release(); d_vblock_needed_.release(); d_active_units_.release(); d_tile_count_.release();
    d_tile_offset_.release(); d_tile_cursor_.release(); d_entries_.release();
    for (auto& kv : textures_) kv.second.texels.release();
    d_textures_.release();
    for (auto& b : d_shadow_order_) b.release();
    for (auto& v : shadow_layers_) for (auto& b : v) b.release();
    d_launch_order_.release(); d_frame_.release(); d_tiles_.release(); d_packed_.release();
    d_hdr_.release(); d_font_.release(); d_srgb_lut_.release(); d_overlay_.release(); d_present_.release();
    for (auto& kv : detile_lists_) kv.second.list.release();
    for (auto& s : staging_) {
        if (s.host) (void)hipHostFree(s.host);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    if (h_counters_) (void)hipHostFree(h_counters_);
    if (h_reports_) (void)hipHostFree(h_reports_);
    for (auto& es : event_sets_)
        for (auto& e : es.ev) if (e) (void)hipEventDestroy(e);
    if (own_stream_ && stream_) (void)hipStreamDestroy(stream_);
    if (geo_stream_) (void)hipStreamDestroy(geo_stream_);
    if (shadow_stream_) (void)hipStreamDestroy(shadow_stream_);
    if (tables_ready_) (void)hipEventDestroy(tables_ready_);
    if (shadows_done_) (void)hipEventDestroy(shadows_done_);
    for (hipEvent_t e : tune_ev_) if (e) (void)hipEventDestroy(e);
    for (int k = 0; k < 2; ++k) {
        if (geo_done_[k]) (void)hipEventDestroy(geo_done_[k]);
        if (shade_done_[k]) (void)hipEventDestroy(shade_done_[k]);
        if (k == 0 && switch_ev_) (void)hipEventDestroy(switch_ev_);
    }
}

void VisualServer::use_device() const { HIP_CHECK(hipSetDevice(device_)); }

// ---- settings ------------------------------------------------------------------------------
// recreate_render_targets — visual_server.rs:621-662: the 3D target is
// (render_size.as_vec2() * render_size_factor).as_uvec2()
void VisualServer::apply_size(uint32_t sw, uint32_t sh, float f) {
    if (!(f > 0.0f) || !(f <= 16.0f)) throw Error(RND_ERR_INVALID, "render_size_factor must be in (0, 16]");
    if (sw == 0 || sh == 0 || sw > kMaxRenderSize || sh > kMaxRenderSize) throw Error(RND_ERR_INVALID, "render size must be in 1..8192");
    const uint32_t w = (uint32_t)((float)sw * f), h = (uint32_t)((float)sh * f);
    if (w == 0 || h == 0 || w > kMaxRenderSize || h > kMaxRenderSize) throw Error(RND_ERR_INVALID, "render size * factor must be in 1..8192");
    if (f != 1.0f && world_ > 1) throw Error(RND_ERR_UNSUPPORTED, "render_size_factor != 1 on a tile partition is not built");
    finish();
    chain_free_streak_ = 0;          // (a changed scene or target: the record chain again until frames show they do not need it)
    tune_reset();
    surface_w_ = sw; surface_h_ = sh; width_ = w; height_ = h; render_size_factor_ = f;
    ext_tiles_ = nullptr; ext_tiles_bytes_ = 0; ext_packed_ = nullptr;
    have_frame_ = false;
    rebuild_tiles();
    if (scaled()) { use_device(); d_present_.ensure((size_t)surface_w_ * surface_h_ * 4); }
}
void VisualServer::set_render_size(uint32_t w, uint32_t h) { apply_size(w, h, render_size_factor_); }
void VisualServer::set_render_size_factor(float f) { apply_size(surface_w_, surface_h_, f); }
void VisualServer::set_msaa(uint32_t n) {
    if (n != 1 && n != 4) throw Error(RND_ERR_INVALID, "sample count must be 1 or 4");
    if (n == msaa_) return;
    finish();
    chain_free_streak_ = 0;          // (a changed scene or target: the record chain again until frames show they do not need it)
    tune_reset();
    msaa_ = n;
    have_frame_ = false;
    rebuild_tiles();    // recreate_render_targets (visual_server.rs:178-181): one visibility plane per sample
}
void VisualServer::set_tone_mapping(uint32_t m) {
    if (m > 1) throw Error(RND_ERR_INVALID, "tone mapping must be 0 (None) or 1 (Reinhard)");
    tone_mapping_ = m;
}

// ---- tiles -----------------------------------------------------------------------------------
std::vector<uint32_t> VisualServer::owned_tiles(uint32_t rank, uint32_t world) const {
    std::vector<uint32_t> out;
    for (uint32_t ty = 0; ty < tiles_y_; ++ty)
        for (uint32_t tx = 0; tx < tiles_x_; ++tx)
            if (world <= 1 || stripe_owner(ty / kOwnerGroup, world) == rank) out.push_back(ty * tiles_x_ + tx);
    return out;
}

void VisualServer::rebuild_tiles() {
    use_device();
    tiles_x_ = (width_ + kTile - 1) / kTile;
    tiles_y_ = (height_ + kTile - 1) / kTile;
    std::vector<uint32_t> own = owned_tiles(rank_, world_);
    n_owned_ = (uint32_t)own.size();
    // Launch order: workgroups b and b+8 share an XCD (round-robin dispatch), so give every XCD one
    // contiguous run of the row-major tile list; neighbouring tiles then share that XCD's L2.
    std::vector<TileRef> order(n_owned_);
    const uint32_t q = n_owned_ / 8, r = n_owned_ % 8;
    for (uint32_t b = 0; b < n_owned_; ++b) {
        uint32_t xcd = b % 8;
        uint32_t idx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
        order[b] = TileRef{own[idx], idx};
    }
    d_launch_order_.ensure(std::max<size_t>(n_owned_, 1));
    if (n_owned_) HIP_CHECK(hipMemcpy(d_launch_order_.ptr, order.data(), n_owned_ * sizeof(TileRef), hipMemcpyHostToDevice));
    const size_t n_tiles = (size_t)tiles_x_ * tiles_y_;
    d_tile_count_.ensure(2 * n_tiles); d_tile_offset_.ensure(n_tiles); d_tile_cursor_.ensure(n_tiles);
    d_vis_[0].ensure(n_tiles * kTilePixels * msaa_); d_vis_[1].ensure(n_tiles * kTilePixels * msaa_);
    d_frame_.ensure((size_t)width_ * height_ * 4);
    if (world_ > 1) d_tiles_.ensure(std::max<size_t>((size_t)n_owned_ * kTilePixels * 4, 4));
}

void VisualServer::set_tile_partition(uint32_t rank, uint32_t world) {
    if (world == 0 || rank >= world) throw Error(RND_ERR_INVALID, "rank must be < world");
    if (world > 1 && scaled()) throw Error(RND_ERR_UNSUPPORTED, "render_size_factor != 1 on a tile partition is not built");
    finish();
    chain_free_streak_ = 0;          // (a changed scene or target: the record chain again until frames show they do not need it)
    tune_reset();
    rank_ = rank; world_ = world;
    ext_tiles_ = nullptr; ext_tiles_bytes_ = 0;
    have_frame_ = false;
    rebuild_tiles();
}
void VisualServer::tile_info(uint32_t& tile_px, uint32_t& n_owned, uint32_t& n_total) const {
    tile_px = kTile; n_owned = n_owned_; n_total = tiles_x_ * tiles_y_;
}

// ---- resources -----------------------------------------------------------------------------
void VisualServer::register_mesh(uint32_t mesh_id, uint32_t n_submeshes, const rnd_submesh_desc* subs) {
    if (n_submeshes && !subs) throw Error(RND_ERR_INVALID, "null submesh array");
    for (uint32_t k = 0; k < n_submeshes; ++k) {
        const rnd_submesh_desc& d = subs[k];
        if ((d.n_vertices && !d.vertices) || (d.n_indices && !d.indices) || d.n_indices % 3)
            throw Error(RND_ERR_INVALID, "bad submesh (null pointer or index count not a multiple of 3)");
        // (the shading kernel addresses a submesh's vertices and indices by 32-bit byte offsets: 32 B per vertex, 4 B per index)
        if ((uint64_t)d.n_vertices * 32u > 0xFFFFFFFFull || (uint64_t)d.n_indices * 4u > 0xFFFFFFFFull)
            throw Error(RND_ERR_INVALID, "submesh exceeds 4 GiB of vertices or indices");
        for (uint32_t i = 0; i < d.n_indices; ++i)
            if (d.indices[i] >= d.n_vertices) throw Error(RND_ERR_INVALID, "vertex index out of range");
    }
    use_device();
    finish();
    chain_free_streak_ = 0;          // (a changed scene or target: the record chain again until frames show they do not need it)
    tune_reset();
    auto old = meshes_.find(mesh_id);
    if (old != meshes_.end()) {
        for (auto& s : old->second.submeshes) { s.vertex_buffer.release(); s.positions.release(); s.index_buffer.release(); s.chunks.release(); }
        meshes_.erase(old);
    }
    RenderMesh mesh;
    for (uint32_t k = 0; k < n_submeshes; ++k) {
        const rnd_submesh_desc& d = subs[k];
        RenderSubmesh s;
        s.vertex_count = d.n_vertices; s.index_count = d.n_indices; s.material = d.material_id;
        s.vertex_buffer.ensure(std::max<size_t>(d.n_vertices, 1));
        s.positions.ensure(std::max<size_t>((size_t)d.n_vertices * 3, 1));
        s.index_buffer.ensure(std::max<size_t>(d.n_indices, 1));
        if (d.n_vertices) HIP_CHECK(hipMemcpyAsync(s.vertex_buffer.ptr, d.vertices, (size_t)d.n_vertices * sizeof(rnd_vertex), hipMemcpyHostToDevice, stream_));
        if (d.n_indices) HIP_CHECK(hipMemcpyAsync(s.index_buffer.ptr, d.indices, (size_t)d.n_indices * 4, hipMemcpyHostToDevice, stream_));
        // one-time AoS -> SoA transpose of the positions for the vertex-transform kernel
        launch_split_positions(s.vertex_buffer.ptr, d.n_vertices, s.positions.ptr, s.positions.ptr + d.n_vertices,
                               s.positions.ptr + 2 * (size_t)d.n_vertices, stream_);
        // per-chunk object-space boxes and vertex ranges for the per-frame work-unit culling
        const uint32_t n_tris = d.n_indices / 3;
        // (both chunkings: the small chunks first, then the large ones, each the union of four small)
        s.n_chunks_small = (n_tris + kChunkTrisSmall - 1) / kChunkTrisSmall;
        s.n_chunks = (n_tris + kChunkTris - 1) / kChunkTris;
        static_assert(kChunkTris % kChunkTrisSmall == 0 && kChunkTrisSmall % 256 == 0, "chunk sizes");
        constexpr uint32_t kPerLarge = kChunkTris / kChunkTrisSmall;
        std::vector<ChunkInfo> infos(s.n_chunks_small + s.n_chunks);
        const float inf = std::numeric_limits<float>::infinity();
        for (uint32_t c = 0; c < s.n_chunks_small; ++c) {
            ChunkInfo ci;
            for (int a = 0; a < 3; ++a) { ci.lo[a] = inf; ci.hi[a] = -inf; }
            ci.vmin = 0xFFFFFFFFu; ci.vmax = 0;
            const size_t i_end = std::min<size_t>((size_t)(c + 1) * kChunkTrisSmall * 3, d.n_indices);
            for (size_t i = (size_t)c * kChunkTrisSmall * 3; i < i_end; ++i) {
                const uint32_t v = d.indices[i];
                ci.vmin = std::min(ci.vmin, v); ci.vmax = std::max(ci.vmax, v);
                const float* p = d.vertices[v].position;
                for (int a = 0; a < 3; ++a) {
                    // a non-finite coordinate opens the box completely: such a chunk is never culled
                    if (!std::isfinite(p[a])) { ci.lo[a] = -inf; ci.hi[a] = inf; }
                    else { ci.lo[a] = std::min(ci.lo[a], p[a]); ci.hi[a] = std::max(ci.hi[a], p[a]); }
                }
            }
            infos[c] = ci;
            ChunkInfo& big = infos[s.n_chunks_small + c / kPerLarge];
            if (c % kPerLarge == 0) big = ci;
            else {
                big.vmin = std::min(big.vmin, ci.vmin); big.vmax = std::max(big.vmax, ci.vmax);
                for (int a = 0; a < 3; ++a) { big.lo[a] = std::min(big.lo[a], ci.lo[a]); big.hi[a] = std::max(big.hi[a], ci.hi[a]); }
            }
            for (int a = 0; a < 3; ++a) {
                if (c == 0) { s.box_lo[a] = ci.lo[a]; s.box_hi[a] = ci.hi[a]; }
                else { s.box_lo[a] = std::min(s.box_lo[a], ci.lo[a]); s.box_hi[a] = std::max(s.box_hi[a], ci.hi[a]); }
            }
        }
        s.chunks.ensure(std::max<size_t>(infos.size(), 1));
        if (!infos.empty()) HIP_CHECK(hipMemcpy(s.chunks.ptr, infos.data(), infos.size() * sizeof(ChunkInfo), hipMemcpyHostToDevice));
        mesh.submeshes.push_back(s);
    }
    HIP_CHECK(hipStreamSynchronize(stream_));   // the caller may free its arrays on return
    meshes_[mesh_id] = std::move(mesh);
}

void VisualServer::register_material(uint32_t id, const Material& m) {
    if (id == RND_NONE || m.billboard_mode > 2) throw Error(RND_ERR_INVALID, "bad material");
    if (m.texture != RND_NONE && !textures_.count(m.texture)) throw Error(RND_ERR_UNKNOWN_ID, "texture id not registered");
    materials_[id] = m;
}
// register_texture / update_texture (visual_server.rs:752-769): an Rgba8UnormSrgb texture with the caller's mip
// chain (backend.rs:358-383).  Stored as uploaded, 4 bytes per texel; the shading kernel decodes at the tap through
// the 512-entry table (sRGB EOTF in f64 rounded once; alpha / 255), so it filters linear values as the hardware
// sampler does.  A texture with any alpha below 255 marks the draw items that use it for the per-fragment alpha test
// (render_mesh.wgsl:91) in the raster stage.
void VisualServer::register_texture(uint32_t id, uint32_t w, uint32_t h, uint32_t mips, const uint8_t* data) {
    if (id == RND_NONE || !w || !h || !mips || !data || w > 16384 || h > 16384) throw Error(RND_ERR_INVALID, "bad texture");
    uint32_t max_levels = 1;
    while ((std::max(w, h) >> max_levels) != 0) ++max_levels;
    if (mips > max_levels) throw Error(RND_ERR_INVALID, "more mip levels than the size allows");
    size_t total = 0;
    for (uint32_t l = 0; l < mips; ++l) total += (size_t)std::max(1u, w >> l) * std::max(1u, h >> l);
    bool opaque = true;
    for (size_t k = 0; k < total; ++k) opaque = opaque && data[4 * k + 3] == 255;
    use_device();
    finish();
    RenderTexture& t = textures_[id];
    const bool is_new = t.mips == 0;
    t.texels.ensure(total * 4);
    HIP_CHECK(hipMemcpy(t.texels.ptr, data, total * 4, hipMemcpyHostToDevice));
    t.width = w; t.height = h; t.mips = mips; t.opaque = opaque;
    if (is_new) t.slot = (uint32_t)textures_.size() - 1;
    upload_texture_table();
}

void VisualServer::upload_texture_table() {
    std::vector<TextureDesc> table(std::max<size_t>(textures_.size(), 1), TextureDesc{nullptr, 0, 0, 0, 0});
    for (const auto& kv : textures_) table.at(kv.second.slot) = TextureDesc{kv.second.texels.ptr, kv.second.width, kv.second.height, kv.second.mips, kv.second.opaque ? 1u : 0u};
    d_textures_.ensure(table.size());
    HIP_CHECK(hipMemcpy(d_textures_.ptr, table.data(), table.size() * sizeof(TextureDesc), hipMemcpyHostToDevice));
}

// ---- per-frame state -------------------------------------------------------------------------
void VisualServer::set_camera(const Mat4& transform, const Camera& camera) {
    projection_ = perspective_lh(camera.vfov, camera.aspect_ratio, camera.near, camera.far);
    view_ = affine_inverse(transform);
    inv_projection_view_ = inverse(projection_ * view_);
    camera_transform_ = transform;
}

void VisualServer::set_light(UniqueNodeId id, const Mat4& transform, const Light& light) {
    if (light.kind > 1) throw Error(RND_ERR_INVALID, "light kind must be 0 (directional) or 1 (point)");
    Light l = light;
    if (l.kind != RND_LIGHT_POINT) l.radius = 0.0f;   // Light::radius().unwrap_or_default()
    RenderLight rl{id, transform, l, {}};
    // create_light / update_light recompute the cascades from the camera current at the call
    // (visual_server.rs:380-381, :445-446); set_camera does not (:221 FIXME)
    // (only directional lights ever read theirs: render_mesh.wgsl:150-152)
    if (l.kind == RND_LIGHT_DIRECTIONAL) shadow_cascade_projviews(inv_projection_view_, transform.axis(2), rl.cascades);
    for (auto& e : lights_)
        if (e.id == id) { e = rl; return; }
    if (lights_.size() >= kMaxLights) throw Error(RND_ERR_INVALID, "too many lights (max 256)");
    lights_.push_back(rl);
}

void VisualServer::set_mesh_instance(UniqueNodeId id, const Mat4& transform, uint32_t mesh_id, uint32_t material_override,
                                     bool casts_shadows) {
    if (!meshes_.count(mesh_id)) throw Error(RND_ERR_UNKNOWN_ID, "mesh id not registered");
    if (material_override != RND_NONE && !materials_.count(material_override)) throw Error(RND_ERR_UNKNOWN_ID, "material id not registered");
    RenderMeshInstance inst{id, transform, mesh_id, material_override, casts_shadows};
    auto found = mesh_instance_index_.find(id);
    if (found != mesh_instance_index_.end()) { mesh_instances_[found->second] = inst; return; }
    mesh_instance_index_.emplace(id, (uint32_t)mesh_instances_.size());
    mesh_instances_.push_back(inst);
}

void VisualServer::set_sprite(UniqueNodeId id, const Mat4& transform, uint32_t texture_id, const float color[4]) {
    if (texture_id != RND_NONE && !textures_.count(texture_id)) throw Error(RND_ERR_UNKNOWN_ID, "texture id not registered");
    uint32_t mid;
    auto it = sprite_materials_.find(id);
    if (it == sprite_materials_.end()) { mid = next_sprite_material_++; sprite_materials_[id] = mid; }
    else mid = it->second;
    Material m;
    std::memcpy(m.base_color, color, 16);
    m.billboard_mode = RND_BILLBOARD_ON; m.unlit = true; m.texture = texture_id;
    materials_[mid] = m;
    set_mesh_instance(id, transform, kQuadMeshId, mid, false);
}

void VisualServer::reset_scene() {
    use_device();
    finish();
    chain_free_streak_ = 0;          // (a changed scene or target: the record chain again until frames show they do not need it)
    tune_reset();
    for (auto it = meshes_.begin(); it != meshes_.end();) {
        if (it->first == kQuadMeshId) { ++it; continue; }
        for (auto& s : it->second.submeshes) { s.vertex_buffer.release(); s.positions.release(); s.index_buffer.release(); s.chunks.release(); }
        it = meshes_.erase(it);
    }
    for (auto& kv : textures_) kv.second.texels.release();
    textures_.clear();
    materials_.clear(); lights_.clear(); mesh_instances_.clear(); mesh_instance_index_.clear(); sprite_materials_.clear();
    fullscreen_ = 0;        // RenderScene::fullscreen_texture (visual_server.rs:903)
    inv_projection_view_ = Mat4::identity();     // RenderScene::default() (visual_server.rs:579-581, :896-898)
}

// ---- 2D overlay state ----------------------------------------------------------------------------
void VisualServer::set_uiboxes(uint32_t n, const rnd_uibox* boxes) {
    if (n && !boxes) throw Error(RND_ERR_INVALID, "null boxes");
    uiboxes_.assign(boxes, boxes + n);
}

void VisualServer::set_font_image(uint32_t w, uint32_t h, const uint8_t* rgba8) {
    if (!w || !h || !rgba8 || w > 16384 || h > 16384) throw Error(RND_ERR_INVALID, "bad font image");
    std::vector<float> texels((size_t)w * h * 4);
    for (size_t k = 0; k < texels.size(); ++k) texels[k] = (float)(rgba8[k] / 255.0);      // create_color_texture_linear: Rgba8Unorm
    use_device();
    finish();
    d_font_.ensure(texels.size());
    HIP_CHECK(hipMemcpy(d_font_.ptr, texels.data(), texels.size() * sizeof(float), hipMemcpyHostToDevice));
    font_w_ = w; font_h_ = h;
}

void VisualServer::reset_texts() { glyphs_.clear(); }

void VisualServer::add_text(const uint8_t* text, uint32_t n, float x, float y, float font_size) {
    if (n && !text) throw Error(RND_ERR_INVALID, "null text");
    for (uint32_t i = 0; i < n; ++i) {      // visual_server.rs:556-570
        Glyph g;
        g.pos[0] = x + (float)i * font_size * 1.1667f * 0.5f; g.pos[1] = y + 0.0f;
        g.scale[0] = font_size * 1.1667f * 0.5f; g.scale[1] = font_size;
        g.glyph = std::min<uint32_t>(text[i], 127u);
        glyphs_.push_back(g);
    }
}

void VisualServer::set_depth_fullscreen_texture() { fullscreen_ = 1; }
void VisualServer::set_shadow_map_fullscreen_texture(UniqueNodeId light) {
    for (const RenderLight& l : lights_)
        if (l.id == light) { fullscreen_ = 2; fullscreen_light_ = light; return; }
    // "warning: no such light registered" and return (visual_server.rs:329-332)
}
void VisualServer::unset_fullscreen_texture() { fullscreen_ = 0; }

// The instanced quads after their vertex stage (uibox.wgsl:30-42, text.wgsl:39-51): corner (cx, cy) in
// {0,1}^2 at pos + c*size, y measured from the top of the surface, to clip space, through the viewport
// transform and onto the sub-pixel grid (raster rule 2), each step rounded in f32 as the shader does.
void VisualServer::build_prims(std::vector<Prim2D>& prims) const {
    const float vw = (float)surface_w_, vh = (float)surface_h_, half_w = 0.5f * vw, half_h = 0.5f * vh;
    auto snap = [](float s) { return (int32_t)std::fmin(std::fmax(std::nearbyint(s * (float)kSub), -1.0e9f), 1.0e9f); };
    auto quad = [&](const float pos[2], const float size[2], Prim2D& q) {
        auto sx = [&](float cx) { const float tp = cx * size[0] + pos[0]; const float clip = (tp / vw) * 2.0f - 1.0f; return snap((clip + 1.0f) * half_w); };
        auto sy = [&](float cy) { const float tp = cy * size[1] + ((vh - pos[1]) - size[1]); const float clip = (tp / vh) * 2.0f - 1.0f; return snap((1.0f - clip) * half_h); };
        q.xa = sx(0.0f); q.xb = sx(1.0f); q.yt = sy(1.0f); q.yb = sy(0.0f);
    };
    prims.reserve(uiboxes_.size() + glyphs_.size());
    for (const rnd_uibox& b : uiboxes_) {
        Prim2D q{};
        quad(b.position, b.size, q);
        for (int k = 0; k < 3; ++k) q.color[k] = std::pow(b.color[k], 2.2f);     // uibox.wgsl:27-28
        q.color[3] = b.color[3];
        q.kind = 0;
        prims.push_back(q);
    }
    for (const Glyph& g : glyphs_) {
        Prim2D q{};
        quad(g.pos, g.scale, q);
        q.kind = 1; q.glyph = g.glyph; q.font_size = g.scale[1];
        prims.push_back(q);
    }
}

void VisualServer::read_overlay_rgba8(uint8_t* dst) {
    finish(); need_frame();
    const size_t bytes = (size_t)surface_w_ * surface_h_ * 4;
    if (!overlay_drawn_) { std::memset(dst, 0, bytes); return; }
    use_device();
    HIP_CHECK(hipMemcpy(dst, d_overlay_.ptr, bytes, hipMemcpyDeviceToHost));
}

// ---- frame -------------------------------------------------------------------------------------
void VisualServer::build_frame(std::vector<DrawItem>& items, std::vector<uint32_t>& vert_bases,
                               std::vector<uint32_t>& tri_bases, std::vector<uint32_t>& unit_bases,
                               std::vector<GpuLight>& lights, std::vector<DrawItem>& shadow_items) {
    const Mat4 pv = projection_ * view_;
    uint64_t vert_base = 0, tri_base = 0, unit_base = 0;
    frame_alpha_test_ = false; frame_textured_ = false;
    bool any_directional = false;
    for (const RenderLight& l : lights_) any_directional = any_directional || l.light.kind == RND_LIGHT_DIRECTIONAL;
    const size_t expect = mesh_instances_.size() + 8;
    items.reserve(expect); vert_bases.reserve(expect + 1); tri_bases.reserve(expect + 1);
    if (any_directional) shadow_items.reserve(expect);
    // for mesh_instance { for submesh } — visual_server.rs:227-245
    for (const RenderMeshInstance& inst : mesh_instances_) {
        const RenderMesh& mesh = meshes_.at(inst.mesh);
        for (const RenderSubmesh& s : mesh.submeshes) {
            uint32_t mid = inst.material_override != RND_NONE ? inst.material_override : s.material;
            Material mat;
            if (mid != RND_NONE) {
                auto it = materials_.find(mid);
                if (it == materials_.end()) throw Error(RND_ERR_UNKNOWN_ID, "material id not registered");
                mat = it->second;
            }
            DrawItem d{};
            d.px = s.positions.ptr; d.py = s.positions.ptr + s.vertex_count; d.pz = s.positions.ptr + 2 * (size_t)s.vertex_count;
            d.verts = s.vertex_buffer.ptr; d.indices = s.index_buffer.ptr;
            d.n_verts = s.vertex_count; d.n_tris = s.index_count / 3;
            d.vert_base = (uint32_t)vert_base; d.tri_base = (uint32_t)tri_base;
            Mat4 world = inst.transform, nrm = inst.transform;
            if (mat.billboard_mode == RND_BILLBOARD_ON) {            // render_mesh.wgsl:58-67
                for (int j = 0; j < 3; ++j) {
                    const Vec4& c = camera_transform_.col[j];
                    world.col[j] = {c.x * 0.4f, c.y * 0.4f, c.z * 0.4f, c.w * 0.4f};
                }
                world.col[3] = inst.transform.col[3];
            } else if (mat.billboard_mode == RND_BILLBOARD_FIXED_SIZE) {   // render_mesh.wgsl:68-79
                const Vec4& o = inst.transform.col[3];
                Vec4 vp = transform(pv, Vec4{o.x, o.y, o.z, 1.0f});
                d.fixed_clip[0] = vp.x / vp.w; d.fixed_clip[1] = vp.y / vp.w; d.fixed_clip[2] = vp.z / vp.w; d.fixed_clip[3] = vp.w / vp.w;
                nrm = Mat4::identity();
            }
            world.to_cols_array(d.world);
            nrm.to_cols_array(d.nrm);
            std::memcpy(d.base_color, mat.base_color, 16);
            d.flags = mat.billboard_mode | (mat.unlit ? kItemUnlit : 0u) | (inst.casts_shadows ? kItemCastsShadows : 0u);
            if (mat.base_color[3] * 1.0f < 0.5f) d.flags |= kItemSkip;   // alpha test, untextured (render_mesh.wgsl:91)
            d.texture = RND_NONE;
            if (mat.texture != RND_NONE) {
                auto tex = textures_.find(mat.texture);
                if (tex == textures_.end()) throw Error(RND_ERR_UNKNOWN_ID, "texture id not registered");
                d.texture = tex->second.slot;
                frame_textured_ = true;
                if (!tex->second.opaque && !(d.flags & kItemSkip)) { d.flags |= kItemAlphaTest; frame_alpha_test_ = true; }
            }
            // small chunks until the frame turns out to be a large one (patched below)
            d.chunks = s.chunks.ptr; d.n_chunks = s.n_chunks_small; d.unit_base = 0;       // (units are numbered below)
            vert_bases.push_back(d.vert_base);
            tri_bases.push_back(d.tri_base);
            vert_base += d.n_verts; tri_base += d.n_tris; unit_base += d.n_chunks;
            if (vert_base >= 0xFFFFFFFFull || tri_base >= 0xFFFFFFFFull || unit_base * kChunkTrisSmall >= 0xFFFFFFF0ull) throw Error(RND_ERR_INVALID, "frame exceeds 2^32 vertices or triangles");
            items.push_back(d);
            // the same geometry as the shadow passes draw it (render_shadow_map.wgsl:25-33: model matrix
            // only, no billboard modes; pipeline3d.rs:268-313: no fragment stage, so no alpha test;
            // instances with casts_shadows = false are skipped)
            if (any_directional) {
                DrawItem sd = d;
                inst.transform.to_cols_array(sd.world);
                sd.flags = inst.casts_shadows ? 0u : kItemSkip;
                shadow_items.push_back(sd);
            }
        }
    }
    vert_bases.push_back((uint32_t)vert_base);
    tri_bases.push_back((uint32_t)tri_base);
    fc_.chunk_tris = kChunkTrisSmall;
    static const uint64_t small_frame_tris = [] { const char* e = std::getenv("RND_SMALL_FRAME_TRIS"); return e ? (uint64_t)std::atoll(e) : (uint64_t)kSmallFrameTris; }();
    if (tri_base > small_frame_tris) {
        // a large frame: the large chunks of every submesh (they follow the small ones in its table)
        fc_.chunk_tris = kChunkTris;
        for (DrawItem& d : items) { d.chunks += d.n_chunks; d.n_chunks = (d.n_tris + kChunkTris - 1) / kChunkTris; }
    }
    // Work units are numbered item by item in PROCESSING order, which is draw order except in frames with depth
    // complexity (early_z_frame_): there the items go near to far (nearest view depth of the item's box), so that the
    // depth test in front of the visibility atomics (k_setup's early-depth variant) finds the near surfaces already
    // resolved when the far ones arrive.  Triangle ids — and with them every tie — stay in draw order.
    const size_t n_items = items.size();
    std::vector<uint32_t>& order = unit_order_;
    order.resize(n_items);
    for (size_t k = 0; k < n_items; ++k) order[k] = (uint32_t)k;
    if (early_z_frame_ && n_items > 1) {
        std::vector<float>& key = unit_order_key_;
        key.resize(n_items);
        size_t k = 0;
        for (const RenderMeshInstance& inst : mesh_instances_) {
            const RenderMesh& mesh = meshes_.at(inst.mesh);
            for (const RenderSubmesh& sub : mesh.submeshes) {
                float nearest = std::numeric_limits<float>::infinity();
                for (int c = 0; c < 8; ++c) {
                    const Vec4 p{(c & 1) ? sub.box_hi[0] : sub.box_lo[0], (c & 2) ? sub.box_hi[1] : sub.box_lo[1], (c & 4) ? sub.box_hi[2] : sub.box_lo[2], 1.0f};
                    const Vec4 v = transform(view_, transform(inst.transform, p));
                    nearest = std::isfinite(v.z) ? std::min(nearest, v.z) : -std::numeric_limits<float>::infinity();
                }
                key[k++] = nearest;
            }
        }
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
    }
    unit_bases.resize(2 * n_items + 1);
    unit_base = 0;
    for (size_t k = 0; k < n_items; ++k) {
        DrawItem& d = items[order[k]];
        d.unit_base = unit_bases[k] = (uint32_t)unit_base;
        unit_bases[n_items + 1 + k] = order[k];
        unit_base += d.n_chunks;
    }
    if (unit_base * fc_.chunk_tris >= 0xFFFFFFF0ull) throw Error(RND_ERR_INVALID, "frame exceeds 2^32 record slots");
    unit_bases[n_items] = (uint32_t)unit_base;
    // the early-depth variant's first pass: the nearest items, up to a tenth of the frame's units (at least one item)
    split_units_ = 0;
    if (early_z_frame_ && n_items > 1) {
        for (size_t k = 0; k + 1 < n_items; ++k) {
            split_units_ = unit_bases[k + 1];
            if ((uint64_t)split_units_ * 10u >= unit_base) break;
        }
    }
    if (any_directional)
        for (size_t k = 0; k < n_items; ++k) { DrawItem& sd = shadow_items[k]; const DrawItem& d = items[k]; sd.chunks = d.chunks; sd.n_chunks = d.n_chunks; sd.unit_base = d.unit_base; }
    fc_.n_units = (uint32_t)unit_base;
    for (const RenderLight& l : lights_) {
        GpuLight g{};
        g.pos[0] = l.transform.col[3].x; g.pos[1] = l.transform.col[3].y; g.pos[2] = l.transform.col[3].z;
        g.dir[0] = l.transform.col[2].x; g.dir[1] = l.transform.col[2].y; g.dir[2] = l.transform.col[2].z;
        g.radius = l.light.radius; g.kind = l.light.kind;
        std::memcpy(g.color, l.light.color, 16);
        g.shadow_index = -1;
        g.inv_radius = l.light.radius > 0.0f ? 1.0f / l.light.radius : 0.0f;
        lights.push_back(g);
    }
    FrameConst& fc = fc_;
    pv.to_cols_array(fc.pv);
    view_.to_cols_array(fc.view);
    fc.cam_pos[0] = camera_transform_.col[3].x; fc.cam_pos[1] = camera_transform_.col[3].y; fc.cam_pos[2] = camera_transform_.col[3].z;
    {
        // (doubles, rounded once: the f32 projection entries are the ones the clip positions were made with)
        // projection_ is a perspective_lh from construction on (SceneUniform defaults), so this always applies
        const double ia = 1.0 / (double)projection_.col[0].x, ib = 1.0 / (double)projection_.col[1].y;
        const Vec4 &cx = camera_transform_.col[0], &cy = camera_transform_.col[1], &cz = camera_transform_.col[2];
        const double q[9] = {cx.x * ia, cx.y * ia, cx.z * ia, cy.x * ib, cy.y * ib, cy.z * ib, cz.x, cz.y, cz.z};
        bool finite = true;
        for (int k = 0; k < 9; ++k) { fc.unproject[k] = (float)q[k]; finite = finite && std::isfinite(fc.unproject[k]); }
        fc.unproject_ok = finite ? 1u : 0u;
    }
    fc.tone = tone_mapping_;
    std::memcpy(fc.ambient, ambient_light_, 16);
    fc.clear_hdr[0] = round_to_f16(0.02f); fc.clear_hdr[1] = round_to_f16(0.04f); fc.clear_hdr[2] = round_to_f16(0.08f);   // Color::GRUE
    fc.width = width_; fc.height = height_; fc.tiles_x = tiles_x_; fc.tiles_y = tiles_y_;
    fc.n_items = (uint32_t)items.size(); fc.n_lights = (uint32_t)lights.size();
    fc.n_tris = (uint32_t)tri_base; fc.n_verts = (uint32_t)vert_base;
    fc.rank = rank_; fc.world = world_;
    fc.ndc_step_x = 2.0 / (double)width_; fc.ndc_step_y = 2.0 / (double)height_;
    fc.ndc_step_xf = (float)fc.ndc_step_x; fc.ndc_step_yf = (float)fc.ndc_step_y;
    fc.n_samples = msaa_;
    fc.n_views = 1; fc.view_h = height_; fc.view_mask = 1u; fc.n_units_view = fc.n_units;      // (multi-view: the shadow passes only)
    fc.sub_dx = fc.sub_dy = 0;        // (MSAA: the four sample positions are evaluated per pixel in ONE geometry pass)
}

// Whether any shadow-casting instance can reach a cascade's clip volume: the bounds of each submesh
// through the instance's transform and the cascade matrix, rejected when all eight corners lie outside
// one plane by a relative margin (the test k_cull applies per chunk, in double here).  A submesh the
// test rejects has every triangle outside the volume, so leaving the pass out draws exactly what
// running it would: nothing.  Scenes with many instances are not examined (the answer is "yes").
bool VisualServer::cascade_has_casters(const Mat4& cascade) const {
    if (mesh_instances_.size() > 32) return true;      // (the test costs the host ~1 us per instance and pass)
    for (const RenderMeshInstance& inst : mesh_instances_) {
        if (!inst.casts_shadows) continue;
        const RenderMesh& mesh = meshes_.at(inst.mesh);
        for (const RenderSubmesh& sub : mesh.submeshes) {
            if (!sub.index_count) continue;
            bool out[6] = {true, true, true, true, true, true};
            bool usable = true;
            for (int c = 0; c < 8 && usable; ++c) {
                const double p[3] = {(c & 1) ? sub.box_hi[0] : sub.box_lo[0], (c & 2) ? sub.box_hi[1] : sub.box_lo[1], (c & 4) ? sub.box_hi[2] : sub.box_lo[2]};
                double w4[4], q[4];
                for (int r = 0; r < 4; ++r) {
                    const Vec4* col = inst.transform.col;
                    const double cr[4] = {r == 0 ? col[0].x : r == 1 ? col[0].y : r == 2 ? col[0].z : col[0].w,
                                          r == 0 ? col[1].x : r == 1 ? col[1].y : r == 2 ? col[1].z : col[1].w,
                                          r == 0 ? col[2].x : r == 1 ? col[2].y : r == 2 ? col[2].z : col[2].w,
                                          r == 0 ? col[3].x : r == 1 ? col[3].y : r == 2 ? col[3].z : col[3].w};
                    w4[r] = cr[0] * p[0] + cr[1] * p[1] + cr[2] * p[2] + cr[3];
                }
                for (int r = 0; r < 4; ++r) {
                    const Vec4* col = cascade.col;
                    const double cr[4] = {r == 0 ? col[0].x : r == 1 ? col[0].y : r == 2 ? col[0].z : col[0].w,
                                          r == 0 ? col[1].x : r == 1 ? col[1].y : r == 2 ? col[1].z : col[1].w,
                                          r == 0 ? col[2].x : r == 1 ? col[2].y : r == 2 ? col[2].z : col[2].w,
                                          r == 0 ? col[3].x : r == 1 ? col[3].y : r == 2 ? col[3].z : col[3].w};
                    q[r] = cr[0] * w4[0] + cr[1] * w4[1] + cr[2] * w4[2] + cr[3] * w4[3];
                }
                if (!(std::isfinite(q[0]) && std::isfinite(q[1]) && std::isfinite(q[2]) && std::isfinite(q[3]))) { usable = false; break; }
                const double e = 1.0e-3 * (std::fabs(q[0]) + std::fabs(q[1]) + std::fabs(q[2]) + std::fabs(q[3])) + 1.0e-30;
                out[0] = out[0] && (q[0] + q[3] < -e); out[1] = out[1] && (q[3] - q[0] < -e);
                out[2] = out[2] && (q[1] + q[3] < -e); out[3] = out[3] && (q[3] - q[1] < -e);
                out[4] = out[4] && (q[2] < -e);        out[5] = out[5] && (q[3] - q[2] < -e);
            }
            if (!usable) return true;
            if (!(out[0] || out[1] || out[2] || out[3] || out[4] || out[5])) return true;
        }
    }
    return false;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// A bump allocator over the static scratch array: sizes are added up first (nothing is dereferenced), the
// frame is rejected before any launch when the sum exceeds the array.
struct ScratchArena {
    uint8_t* base; size_t size; size_t used = 0;
    template <class T> T* take(size_t n) {
        const size_t off = (used + 255) / 256 * 256;
        used = off + std::max<size_t>(n, 1) * sizeof(T);
        return reinterpret_cast<T*>(base + off);
    }
};

// This frame's launch mode: true = geometry on the internal stream beside the previous frame's shading, false =
// everything on the context's stream.  Frames with 60 us of shading or more overlap (measured: C3 +6 %, C4 +4 %).  Below
// that the hand-over between the streams (two cross-queue dependencies per frame) can cost more than the overlap gains,
// and which way it goes does not follow from the shading time (C2, 27 us: 13 760 frames/s overlapped, 12 950 serial; the
// share of one rank of eight of C3, 26 us: 12 700 / 14 060; C1', 19 us: 6650 / 7030), so both are measured: kTuneWindow
// frames each after kTuneSettle frames to settle, between two events on the context's stream, no synchronization (the
// result is looked at when the last event has completed).  A timed frame or a finish() inside a window voids it.
bool VisualServer::tune_overlap(bool timed) {
    const bool guess = !(have_timings_ && timings_.shade_ms < 0.03f);
    if (!tune_enabled_ || !overlap_) return guess;
    if (have_timings_ && timings_.shade_ms >= 0.06f) { tune_reset(); return true; }
    const bool fallback = tune_decided_ ? tune_choice_ : guess;
    if (timed) {
        if (tune_state_ == kTuneProbeOverlap || tune_state_ == kTuneProbeSerial) tune_state_ = kTuneIdle;
        tune_mark_ = -1;
        return fallback;
    }
    switch (tune_state_) {
    case kTuneIdle:
        tune_state_ = kTuneProbeOverlap; tune_i_ = 0;
        [[fallthrough]];
    case kTuneProbeOverlap:
    case kTuneProbeSerial: {
        const bool mode = tune_state_ == kTuneProbeOverlap;
        const int base = mode ? 0 : 2;
        if (tune_i_ == kTuneSettle) tune_mark_ = base;
        if (tune_i_ == kTuneSettle + kTuneWindow) {
            tune_mark_ = base + 1;
            tune_state_ = mode ? kTuneProbeSerial : kTuneWait;
            tune_i_ = 0;
        } else {
            ++tune_i_;
        }
        return mode;
    }
    case kTuneWait:
        if (hipEventQuery(tune_ev_[3]) == hipSuccess) {
            if (hipEventElapsedTime(&tune_ms_[0], tune_ev_[0], tune_ev_[1]) == hipSuccess &&
                hipEventElapsedTime(&tune_ms_[1], tune_ev_[2], tune_ev_[3]) == hipSuccess) {
                // (the mode in use stays unless the other one is 2 % faster)
                const bool was = tune_decided_ ? tune_choice_ : guess;
                tune_choice_ = was ? !(tune_ms_[1] < 0.98f * tune_ms_[0]) : (tune_ms_[0] < 0.98f * tune_ms_[1]);
                tune_decided_ = true;
                static const bool dbg = std::getenv("RND_TUNE_DEBUG") != nullptr;
                if (dbg) std::fprintf(stderr, "rnd tune: %u frames overlapped %.3f ms, serial %.3f ms -> %s\n", kTuneWindow, tune_ms_[0], tune_ms_[1], tune_choice_ ? "overlap" : "serial");
            }
            tune_state_ = kTuneHolding; tune_i_ = 0;
        }
        (void)hipGetLastError();        // (hipErrorNotReady is not an error here)
        return fallback;
    case kTuneHolding:
        if (++tune_i_ >= kTuneHold) tune_state_ = kTuneIdle;
        return fallback;
    }
    return fallback;
}

void VisualServer::enqueue_frame(bool want_hdr, uint8_t* noalloc_out, uint64_t noalloc_bytes) {
    const bool noalloc = noalloc_out != nullptr;
    static const bool prof = std::getenv("RND_HOST_PROFILE") != nullptr;
    static double acc[6]; static uint32_t acc_n;
    static double sub[6];       // inside the launches: waits, geometry launches, hand-over events, shading launch, end events
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t_start = now(); auto t_prev = t_start;
    auto lap = [&](int k) { if (prof) { auto t = now(); acc[k] += std::chrono::duration<double, std::micro>(t - t_prev).count(); t_prev = t; } };
    use_device();
    assembled_frame_ = false;
    poll_reports();                     // overflow reports of the frames in flight (no wait): streams grow before this frame is laid out
    if (scaled()) want_hdr = true;      // the present pass samples the HDR target
    const uint32_t p = parity_ ^ 1u;    // this frame's copy of the buffers that geometry writes and shading reads
    // Overlap pays when there is shading to hide the geometry chain behind: the hand-over between the two
    // streams costs ~25 us per frame (measured: C3 +6 %, C4 +4 %, C2 +2 %; C1', 20 us of shading, -8 %),
    // so frames whose last measured shading time is below 30 us stay on one stream.
    const bool timed_frame = force_timed_ > 0 || frame_counter_ < 4 || frame_counter_ % kTimingStride == 0;
    const bool overlap = overlap_ && !noalloc && tune_overlap(timed_frame);
    hipStream_t const gs = overlap ? geo_stream_ : stream_;      // the geometry phase's stream
    std::vector<DrawItem>& items = items_;
    std::vector<uint32_t>&vert_bases = vert_bases_, &tri_bases = tri_bases_, &unit_bases = unit_bases_;
    std::vector<GpuLight>& lights = gpu_lights_;
    std::vector<DrawItem>& shadow_items = shadow_items_;
    items.clear(); vert_bases.clear(); tri_bases.clear(); unit_bases.clear(); lights.clear(); shadow_items.clear();
    // A no-allocation frame can be refused (what the scratch array cannot hold); a refused frame leaves the context as it
    // was: what can be told beforehand is checked before anything changes, and what laying the frame out changes is put
    // back when a later check refuses it.
    struct Restore {
        VisualServer& vs; bool armed;
        FrameConst fc; Capacities cap; uint32_t clip_records; bool alpha, textured, early; uint32_t split;
        ~Restore() {
            if (!armed) return;
            vs.fc_ = fc; vs.cap_ = cap; vs.clip_records_ = clip_records; vs.frame_alpha_test_ = alpha; vs.frame_textured_ = textured;
            vs.early_z_frame_ = early; vs.split_units_ = split;
        }
    } restore{*this, noalloc, fc_, cap_, clip_records_, frame_alpha_test_, frame_textured_, early_z_frame_, split_units_};
    if (noalloc) {
        for (const RenderLight& l : lights_)
            if (l.light.kind == RND_LIGHT_DIRECTIONAL) throw Error(RND_ERR_UNSUPPORTED, "rnd_render_noalloc: directional lights need 134 MB of shadow cascades each, beyond the scratch array");
        if (scaled()) throw Error(RND_ERR_UNSUPPORTED, "rnd_render_noalloc: render_size_factor != 1 needs an HDR plane");
        if (fullscreen_ != 0 || !uiboxes_.empty() || !glyphs_.empty()) throw Error(RND_ERR_UNSUPPORTED, "rnd_render_noalloc: the 2D overlay is not available here");
        if (want_packed_ && d_packed_.count < (size_t)width_ * height_) throw Error(RND_ERR_UNSUPPORTED, "rnd_render_noalloc: the packed frame has not been allocated for this size");
        const uint64_t need_out = world_ > 1 ? (uint64_t)n_owned_ * kTilePixels * 4 : (uint64_t)width_ * height_ * 4;
        if (noalloc_bytes < need_out) throw Error(RND_ERR_INVALID, "rnd_render_noalloc: output buffer too small");
        // the array may still be another context's
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        VisualServer* const holder = g_scratch_owner[device_];
        if (holder && holder != this && hipEventQuery(g_scratch_busy[device_]) == hipErrorNotReady)
            throw Error(RND_ERR_INVALID, "rnd_render_noalloc: another context's no-allocation frame is still in flight on this GPU (one scratch array per GPU)");
        (void)hipGetLastError();
    }
    {
        const char* const e = std::getenv("RND_EARLY_DEPTH");
        early_z_frame_ = e && *e ? std::atoi(e) != 0 : early_z_state_;
    }
    build_frame(items, vert_bases, tri_bases, unit_bases, lights, shadow_items);
    // every directional light owns a cascade set (the reference also fills maps for point lights,
    // which its shader never reads: visual_server.rs:358-436, render_mesh.wgsl:150-152)
    std::vector<uint32_t> shadowed;
    for (uint32_t i = 0; i < lights.size(); ++i)
        if (lights[i].kind == RND_LIGHT_DIRECTIONAL) { lights[i].shadow_index = (int32_t)shadowed.size(); shadowed.push_back(i); }
    const uint32_t n_sets = fc_.n_tris ? (uint32_t)shadowed.size() : 0u;
    if (!n_sets) for (auto& l : lights) l.shadow_index = -1;
    // every shadowed light is ONE multi-view geometry pass: its four cascades are the views, stacked vertically in a
    // kShadowSize x 4*kShadowSize target (the four contiguous layers of the light's maps)
    const uint32_t n_passes = n_sets;
    constexpr uint32_t kShadowTiles = (kShadowSize / kTile) * (kShadowSize / kTile);
    const uint32_t n_views_max = n_sets ? kShadowCascades : 1u;        // views of the frame's widest pass (buffer sizes)
    static_assert(kShadowCascades <= kMaxViews && kShadowCascades * kShadowSize <= kMaxRenderSize, "the cascades must stack into one render target");
    const bool overlay = fullscreen_ != 0 || !uiboxes_.empty() || !glyphs_.empty();
    // Shadow passes on their own stream beside the main pass's geometry (second buffer set, below)?  Measured: C1' 720p
    // 6990 -> 9560 frames/s, at 4K 5490 -> 7520, C2 + sun 5300 -> 5850: chains of small launches run side by side; C3 +
    // sun + texture, 0.27 ms of shadow pass: 1505 -> 1476, two heavy passes only share the machine — so only while the last
    // measured shadow pass took less than 0.2 ms.
    const bool shadows_beside_frame = shadow_concurrent_ && n_sets != 0 && !noalloc && !(have_timings_ && timings_.shadow_ms >= 0.2f);
    if (noalloc && n_sets) throw Error(RND_ERR_UNSUPPORTED, "rnd_render_noalloc: directional lights need shadow cascades beyond the scratch array");
    static_assert(kPinnedCounterSets >= 1, "a frame without shadow passes has one counter set, pinned at creation");
    std::vector<ShadowSet> sets(n_sets);
    if (n_sets) {
        if (shadow_layers_[p].size() < n_sets) shadow_layers_[p].resize(n_sets);
        if (shadow_layer_clear_[p].size() < n_sets) shadow_layer_clear_[p].resize(n_sets, std::array<bool, 4>{false, false, false, false});
        for (uint32_t k = 0; k < n_sets; ++k) {
            shadow_layers_[p][k].ensure((size_t)kShadowCascades * kShadowTiles * kTilePixels);
            const RenderLight& rl = lights_[shadowed[k]];
            for (int c = 0; c < 3; ++c) rl.cascades[c].to_cols_array(sets[k].world_to_light[c]);
            for (uint32_t c = 0; c < kShadowCascades; ++c) sets[k].layers[c] = shadow_layers_[p][k].ptr + (size_t)c * kShadowTiles * kTilePixels;
        }
        const size_t need = std::max<size_t>((size_t)tiles_x_ * tiles_y_, (size_t)kShadowCascades * kShadowTiles);
        d_tile_count_.ensure(2 * need); d_tile_offset_.ensure(need); d_tile_cursor_.ensure(need);
    }
    // (members that describe "the last enqueued frame" change only after the last point where this one can be rejected)
    const uint32_t n_counter_sets = 1 + n_passes;
    if (!noalloc) d_counters_[p].ensure(n_counter_sets);
    if (h_counter_sets_ < n_counter_sets) {
        if (frame_pending_) HIP_CHECK(hipStreamSynchronize(stream_));
        if (h_counters_) HIP_CHECK(hipHostFree(h_counters_));
        h_counters_ = nullptr;
        HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h_counters_), n_counter_sets * sizeof(FrameCounters), hipHostMallocDefault));
        h_counter_sets_ = n_counter_sets;
    }
    const FrameConst& fc = fc_;
    const size_t n_tiles = (size_t)tiles_x_ * tiles_y_;

    // capacities
    if ((uint64_t)fc.n_units * n_views_max * fc.chunk_tris >= 0xFFFFFFF0ull) throw Error(RND_ERR_INVALID, "frame exceeds 2^32 record slots");
    cap_.setup_slots = fc.n_units * n_views_max * fc.chunk_tris;
    // (RND_TEST_SMALL_STREAMS: floors so small that ordinary test scenes overflow them, for the tests of that path)
    const uint32_t floor_clip_records = test_small_streams_ ? 8u : fc.n_tris / 8 + 4096;
    const uint32_t floor_clip_queue = test_small_streams_ ? 16u : std::max<uint32_t>(fc.n_tris / 8, 65536);
    clip_records_ = std::max<uint32_t>(clip_records_, floor_clip_records);
    cap_.records = (uint32_t)std::min<uint64_t>((uint64_t)cap_.setup_slots + clip_records_, 0xFFFFFFF0ull);
    if (cap_.clip_queue < floor_clip_queue) cap_.clip_queue = floor_clip_queue;
    const uint64_t want_entries = test_small_streams_ ? 64u : std::max<uint64_t>(4ull * cap_.records + 16ull * n_tiles, 1u << 20);
    if (cap_.entries < want_entries) cap_.entries = (uint32_t)std::min<uint64_t>(want_entries, 0xFFFFFFF0ull);
    // the per-frame intermediate streams: hipMalloc'ed buffers that grow on demand, or — for the no-allocation
    // frame — slices of the static scratch array
    ScratchArena arena{scratch_, kScratchBytes};
    struct {
        float4* clip; SnapVert* snap; SetupRecord* records; uint32_t *tilewords, *clip_queue, *chunk_count, *active_units, *vblock_needed, *entries,
            *wide_list, *tile_slices, *tile_count, *tile_offset, *tile_cursor;
        uint2* raster_extra; FrameCounters* counters; uint64_t* vis; float* hiz; unsigned long long* stripes; uint4* unit_bounds; uint2* far_units;
    } bp{};
    const size_t n_vblocks = ((size_t)fc.n_verts / kVertexBlock + 2) * std::max<size_t>(n_views_max, early_z_frame_ ? 2 : 1);       // (early-depth frames: a second row of flags)
    const size_t n_unit_words = std::max<size_t>(2 * (size_t)fc.n_units * n_views_max, 2);
    const size_t n_count_words = std::max<size_t>((size_t)fc.n_units * n_views_max, 1);          // per unit: record slots used
    const size_t n_view_verts = std::max<size_t>((size_t)fc.n_verts * n_views_max, 1);
    // Whether K1 keeps the clip positions for the shading kernel or that kernel makes them again from the corners' vertices
    // (kernels.hip, corner_clip: the same bits either way).  Storing costs K1 16 of its 44 bytes per vertex (1.45 us per
    // million vertices) and two buffers of 16 B per vertex; re-deriving costs the shading kernel 36 instructions per
    // corner (18 us on the 8.3 M pixels of a 4K frame).  Measured: C3 (5 M vertices at 4K) 3370 frames/s storing, 3330
    // re-deriving; C5 (50 M vertices at 4K) 797 storing, 890 re-deriving.
    // (RND_STORE_CLIP = 0 / 1 forces one; read per frame, so that one process can test both)
    const char* const store_clip_env = std::getenv("RND_STORE_CLIP");
    const bool store_clip = store_clip_env && *store_clip_env ? std::atoi(store_clip_env) != 0
                                                : (uint64_t)fc.n_verts * 2u <= 3ull * (uint64_t)width_ * height_ * msaa_;
    // (16 B per vertex behind a 32-bit byte offset: a frame of 2^28 vertices or more re-derives whatever was asked for)
    const bool store_clip_fits = fc.n_verts < (1u << 28);
    if (noalloc) {
        bp.clip = store_clip && store_clip_fits ? arena.take<float4>(std::max<size_t>(fc.n_verts, 1)) : nullptr; bp.snap = arena.take<SnapVert>(n_view_verts);
        bp.records = arena.take<SetupRecord>(cap_.records); bp.tilewords = arena.take<uint32_t>(2 * (size_t)cap_.records);
        bp.clip_queue = arena.take<uint32_t>(2 * (size_t)cap_.clip_queue);      // triangle ids, then their views
        bp.chunk_count = arena.take<uint32_t>(n_count_words);
        bp.active_units = arena.take<uint32_t>(n_unit_words);
        bp.vblock_needed = arena.take<uint32_t>(n_vblocks);
        bp.entries = arena.take<uint32_t>(cap_.entries);
        bp.wide_list = arena.take<uint32_t>(cap_.wide_list);
        bp.tile_slices = arena.take<uint32_t>(n_tiles);
        bp.raster_extra = arena.take<uint2>(kRasterExtraCap);
        bp.tile_count = arena.take<uint32_t>(2 * n_tiles); bp.tile_offset = arena.take<uint32_t>(n_tiles); bp.tile_cursor = arena.take<uint32_t>(n_tiles);
        bp.counters = arena.take<FrameCounters>(n_counter_sets);
        bp.stripes = arena.take<unsigned long long>((size_t)n_counter_sets * kStatStripes * kStatStride);
        bp.vis = arena.take<uint64_t>(n_tiles * kTilePixels * msaa_);
        bp.hiz = early_z_frame_ ? arena.take<float>(n_tiles * 17) : nullptr;
        bp.unit_bounds = early_z_frame_ ? arena.take<uint4>(std::max<size_t>(fc.n_units, 1)) : nullptr;
        bp.far_units = early_z_frame_ ? arena.take<uint2>(std::max<size_t>(fc.n_units, 1)) : nullptr;
    } else {
        if (store_clip && store_clip_fits) d_clip_[p].ensure(std::max<size_t>(fc.n_verts, 1));        // (the main pass's only: shading reads it)
        d_snap_.ensure(n_view_verts);
        d_records_.ensure(cap_.records); d_tilewords_.ensure(2 * (size_t)cap_.records);
        d_clip_queue_.ensure(2 * (size_t)cap_.clip_queue);
        d_chunk_count_.ensure(n_count_words);
        d_active_units_.ensure(n_unit_words);
        d_vblock_needed_.ensure(n_vblocks);
        d_entries_.ensure(cap_.entries);
        d_wide_list_.ensure(cap_.wide_list);
        d_tile_slices_.ensure(std::max<size_t>(n_tiles, n_sets ? (size_t)kShadowCascades * kShadowTiles : 0));
        d_raster_extra_.ensure(kRasterExtraCap);
        if (shadows_beside_frame) {
            GeoSet& g = shadow_geo_;
            g.snap.ensure(n_view_verts); g.records.ensure(cap_.records); g.tilewords.ensure(2 * (size_t)cap_.records);
            g.clip_queue.ensure(2 * (size_t)cap_.clip_queue); g.chunk_count.ensure(n_count_words); g.active_units.ensure(n_unit_words);
            g.vblock_needed.ensure(n_vblocks); g.entries.ensure(cap_.entries); g.wide_list.ensure(cap_.wide_list);
            const size_t need = (size_t)kShadowCascades * kShadowTiles;
            g.tile_slices.ensure(need); g.raster_extra.ensure(kRasterExtraCap);
            g.tile_count.ensure(2 * need); g.tile_offset.ensure(need); g.tile_cursor.ensure(need);
        }
        if (world_ > 1) d_tiles_.ensure(std::max<size_t>((size_t)n_owned_ * kTilePixels * 4, 4));
        if (want_packed_) d_packed_.ensure((size_t)width_ * height_);
        if (want_hdr) d_hdr_.ensure((size_t)width_ * height_ * 3);
        bp.clip = store_clip && store_clip_fits ? d_clip_[p].ptr : nullptr; bp.snap = d_snap_.ptr; bp.records = d_records_.ptr; bp.tilewords = d_tilewords_.ptr; bp.clip_queue = d_clip_queue_.ptr;
        bp.chunk_count = d_chunk_count_.ptr; bp.active_units = d_active_units_.ptr; bp.vblock_needed = d_vblock_needed_.ptr; bp.entries = d_entries_.ptr;
        bp.wide_list = d_wide_list_.ptr; bp.tile_slices = d_tile_slices_.ptr; bp.raster_extra = d_raster_extra_.ptr;
        bp.tile_count = d_tile_count_.ptr; bp.tile_offset = d_tile_offset_.ptr; bp.tile_cursor = d_tile_cursor_.ptr;
        bp.counters = d_counters_[p].ptr; bp.vis = d_vis_[p].ptr;
        d_stat_stripes_[p].ensure((size_t)n_counter_sets * kStatStripes * kStatStride);
        bp.stripes = reinterpret_cast<unsigned long long*>(d_stat_stripes_[p].ptr);
        if (early_z_frame_) { d_hiz_.ensure(n_tiles * 17); d_unit_bounds_.ensure(std::max<size_t>(fc.n_units, 1)); d_far_units_.ensure(std::max<size_t>(fc.n_units, 1)); }
        bp.hiz = d_hiz_.ptr;
        bp.unit_bounds = early_z_frame_ ? d_unit_bounds_.ptr : nullptr;
        bp.far_units = early_z_frame_ ? d_far_units_.ptr : nullptr;
    }

    // stage the per-frame tables in pinned memory and upload them
    const size_t off_items = 0;
    const size_t off_vb = align_up(off_items + items.size() * sizeof(DrawItem), 256);
    const size_t off_tb = align_up(off_vb + vert_bases.size() * 4, 256);
    const size_t off_ub = align_up(off_tb + tri_bases.size() * 4, 256);
    const size_t off_li = align_up(off_ub + unit_bases.size() * 4, 256);
    const size_t off_si = align_up(off_li + lights.size() * sizeof(GpuLight), 256);
    const size_t off_ss = align_up(off_si + (n_sets ? shadow_items.size() * sizeof(DrawItem) : 0), 256);
    // the constants of every pass of the frame: [0] main pass, then one per shadowed light
    std::vector<FrameConst>& consts = consts_;
    consts.assign(1 + n_passes, fc);
    // Which cascades of which light are drawn / cleared this frame.  A cascade no caster can reach stays cleared: it
    // is left out of the pass (view_mask), and its layer is only cleared again if it was drawn into since.
    std::vector<uint32_t> draw_mask(n_sets, 0u), clear_mask(n_sets, 0u);
    for (uint32_t k = 0; k < n_sets; ++k) {
        for (uint32_t c = 0; c < kShadowCascades; ++c) {
            const bool casters = cascade_has_casters(lights_[shadowed[k]].cascades[c]);
            if (casters) draw_mask[k] |= 1u << c;
            if (casters || !shadow_layer_clear_[p][k][c]) clear_mask[k] |= 1u << c;
        }
        FrameConst& sfc = consts[1 + k];
        const Mat4* cas = lights_[shadowed[k]].cascades;
        cas[0].to_cols_array(sfc.pv);
        for (uint32_t c = 1; c < kShadowCascades; ++c) cas[c].to_cols_array(sfc.pv_more[c - 1]);
        sfc.width = kShadowSize; sfc.height = kShadowCascades * kShadowSize;
        sfc.tiles_x = kShadowSize / kTile; sfc.tiles_y = kShadowCascades * kShadowSize / kTile;
        sfc.n_views = kShadowCascades; sfc.view_h = kShadowSize; sfc.view_mask = draw_mask[k];
        sfc.n_units_view = fc.n_units; sfc.n_units = fc.n_units * kShadowCascades;
        sfc.rank = 0; sfc.world = 1; sfc.n_lights = 0;
        sfc.n_samples = 1; sfc.sub_dx = sfc.sub_dy = 0;      // shadow maps are single-sampled (visual_server.rs:371)
    }
    constexpr size_t kConstSlot = (sizeof(FrameConst) + 255) / 256 * 256;
    const size_t off_fc = align_up(off_ss + n_sets * sizeof(ShadowSet), 256);
    // the 2D pass's quads, when anything 2D is set
    std::vector<Prim2D> prims;
    if (overlay) build_prims(prims);
    const size_t off_pr = align_up(off_fc + consts.size() * kConstSlot, 256);
    const size_t total = align_up(off_pr + prims.size() * sizeof(Prim2D) + 256, 256);
    lap(0);
    Staging& st = staging_[staging_next_];
    size_t tables_off = 0;
    if (st.bytes < total && noalloc) throw Error(RND_ERR_INVALID, "rnd_render_noalloc: the frame's tables (draw items, lights) exceed the 1 MiB staging block allocated at rnd_create");
    if (noalloc) {
        tables_off = static_cast<size_t>(arena.take<uint8_t>(total) - scratch_);       // the device copy of the tables
        if (arena.used > arena.size) {
            char msg[200];
            std::snprintf(msg, sizeof msg, "rnd_render_noalloc: this frame's intermediate streams need %zu bytes, the static scratch array holds %zu "
                          "(roughly 1 M triangles at 1080p)", arena.used, arena.size);
            throw Error(RND_ERR_INVALID, msg);
        }
    }
    // nothing below rejects the frame
    restore.armed = false;
    if (noalloc) {
        // the array changes hands: the previous holder's last frame (its visibility words, its counters) is gone
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        VisualServer* const holder = g_scratch_owner[device_];
        if (holder && holder != this && holder->last_noalloc_) { holder->have_frame_ = false; holder->frame_pending_ = false; }
        g_scratch_owner[device_] = this;
    }
    parity_ = p;
    n_counter_sets_ = n_counter_sets;
    counter_set_ran_.assign(n_counter_sets_, 0);
    staging_next_ = (staging_next_ + 1) % 3;
    if (st.in_flight) { HIP_CHECK(hipEventSynchronize(st.done)); st.in_flight = false; }
    if (st.bytes < total) {
        if (st.host) HIP_CHECK(hipHostFree(st.host));
        st.host = nullptr;
        // mapped + portable: the first clear kernel reads this block itself, from whichever device the context is on
        HIP_CHECK(hipHostMalloc(&st.host, total + total / 2, hipHostMallocMapped | hipHostMallocPortable));
        HIP_CHECK(hipHostGetDevicePointer(&st.host_dev, st.host, 0));
        st.bytes = total + total / 2;
    }
    lap(1);
    char* h = static_cast<char*>(st.host);
    if (!items.empty()) std::memcpy(h + off_items, items.data(), items.size() * sizeof(DrawItem));
    std::memcpy(h + off_vb, vert_bases.data(), vert_bases.size() * 4);
    std::memcpy(h + off_tb, tri_bases.data(), tri_bases.size() * 4);
    std::memcpy(h + off_ub, unit_bases.data(), unit_bases.size() * 4);
    if (!lights.empty()) std::memcpy(h + off_li, lights.data(), lights.size() * sizeof(GpuLight));
    if (n_sets) {
        std::memcpy(h + off_si, shadow_items.data(), shadow_items.size() * sizeof(DrawItem));
        std::memcpy(h + off_ss, sets.data(), n_sets * sizeof(ShadowSet));
    }
    for (size_t k = 0; k < consts.size(); ++k) std::memcpy(h + off_fc + k * kConstSlot, &consts[k], sizeof(FrameConst));
    if (!prims.empty()) std::memcpy(h + off_pr, prims.data(), prims.size() * sizeof(Prim2D));
    if (overlay) d_overlay_.ensure((size_t)surface_w_ * surface_h_);
    overlay_drawn_ = overlay;
    // one device block with the same layout; the frame's first clear kernel copies it over from the
    // pinned block (no hipMemcpyAsync: each one costs the host ~8 us and the GPU a blit kernel)
    uint8_t* tb;
    if (noalloc) {
        tb = scratch_ + tables_off;
    } else {
        d_tables_[p].ensure(total + total / 2);
        tb = d_tables_[p].ptr;
    }

    lap(2);
    FrameBuffers fb{};
    auto const_slot = [&](size_t k) { return reinterpret_cast<const FrameConst*>(tb + off_fc + k * kConstSlot); };
    fb.frame_const = const_slot(0);
    fb.items = reinterpret_cast<const DrawItem*>(tb + off_items); fb.vert_bases = reinterpret_cast<const uint32_t*>(tb + off_vb);
    fb.tri_bases = reinterpret_cast<const uint32_t*>(tb + off_tb); fb.lights = reinterpret_cast<const GpuLight*>(tb + off_li);
    fb.textures = d_textures_.ptr; fb.srgb_lut = d_srgb_lut_.ptr;
    fb.alpha_test = frame_alpha_test_ ? 1u : 0u;
    fb.has_textures = frame_textured_ ? 1u : 0u;
    fb.shadow_sets = n_sets ? reinterpret_cast<const ShadowSet*>(tb + off_ss) : nullptr;      // null selects the shading kernel without the lookup
    fb.n_shadow_sets = n_sets;
    // measured: 16 -> 32 takes C4 (bboxes of ~20 pixel centres) from 230 to 285 frames/s by keeping it out of the
    // record/bin/raster path; C3 and C5 do not move; 64 gains nothing more and costs C1
    static const uint32_t direct_pixels = [] { const char* e = std::getenv("RND_DIRECT_PIXELS"); int v = e ? std::atoi(e) : (int)kDirectPixels; return (uint32_t)std::min(std::max(v, 0), 64); }();
    fb.direct_pixels = direct_pixels;       // (64 in early-depth frames: below, once it is known whether the frame runs that variant)
    fb.clip = bp.clip; fb.snap = bp.snap; fb.records = bp.records; fb.tilewords = reinterpret_cast<uint2*>(bp.tilewords); fb.clip_queue = bp.clip_queue;
    fb.chunk_count = bp.chunk_count;
    fb.unit_bases = reinterpret_cast<const uint32_t*>(tb + off_ub); fb.active_units = reinterpret_cast<uint2*>(bp.active_units); fb.vblock_needed = bp.vblock_needed;
    fb.tile_count = bp.tile_count; fb.tile_offset = bp.tile_offset; fb.tile_cursor = bp.tile_cursor;
    fb.wide_list = bp.wide_list; fb.tile_slices = bp.tile_slices; fb.raster_extra = bp.raster_extra;
    fb.entries = bp.entries; fb.counters = bp.counters; fb.main_counters = bp.counters; fb.stat_stripes = bp.stripes;
    // without the record chain: after a run of frames that needed none of it (never the no-allocation frame, whose
    // statistics nobody could sum: its chunk counts live in the shared array)
    frame_no_chain_ = allow_no_chain_ && !noalloc && full_chain_frames_ == 0 && chain_free_streak_ >= kNoChainStreak;
    if (full_chain_frames_) --full_chain_frames_;
    fb.no_chain = frame_no_chain_ ? 1u : 0u;
    // (the early-depth variant exists for the plain main pass only: k_cull parts near and far units only when it runs)
    const bool early_z_used = early_z_frame_ && !frame_alpha_test_ && msaa_ == 1;
    fb.early_z = early_z_used ? 1u : 0u;
    // In an early-depth frame every triangle the first pass leaves to the record path is a hole in k_hiz's picture (its pixels
    // are still clear when k_hiz looks), and one hole in a far unit's rect keeps the unit: at C4 the 0.1 % of the nearest
    // layer's triangles with bboxes of 33..64 pixel centres kept 14 004 of 43 947 hidden units alive (745 frames/s; with
    // them rasterized at setup too: 3 units, 1120 frames/s, and no record left for the chain).
    static const bool direct_env = std::getenv("RND_DIRECT_PIXELS") != nullptr;
    if (early_z_used && !direct_env) fb.direct_pixels = 64u;
    fb.split_units = split_units_;
    fb.setup_pass = early_z_used && split_units_ != 0u && split_units_ < fc.n_units ? 1u : 0u;
    fb.hiz = bp.hiz;
    fb.bbox_px_hint_x16 = bbox_px_hint_x16_;
    fb.unit_bounds = bp.unit_bounds;
    fb.far_units = bp.far_units;
    fb.launch_order = d_launch_order_.ptr; fb.n_owned = n_owned_;
    fb.dims = ShadeDims{fc.n_lights, fc.n_items, fc.width, fc.height, fc.tiles_x,
                        fc.tiles_x > 1u ? (uint32_t)((1ull << 32) / fc.tiles_x) + 1u : 0u, fc.rank, fc.world};
    fb.vis = bp.vis;
    fb.vis_stride = (uint64_t)tiles_x_ * tiles_y_ * kTilePixels;
    fb.frame_rgba8 = (world_ > 1 || scaled()) ? nullptr : (noalloc ? noalloc_out : d_frame_.ptr);
    fb.tiles_rgba8 = world_ > 1 ? (noalloc ? noalloc_out : ext_tiles_ ? ext_tiles_ : d_tiles_.ptr) : nullptr;
    fb.all_counters = bp.counters; fb.n_counter_sets = n_counter_sets;
    fb.ran_mask = ~0ull; fb.report = nullptr; fb.report_tag = 0;
    last_report_slot_ = kReportSlots;
    {
        const uint32_t slot = report_next_;
        report_next_ = (report_next_ + 1) % kReportSlots;
        if (report_expect_[slot]) {         // kReportSlots frames in flight: that frame's report must be seen before its slot is reused
            HIP_CHECK(hipStreamSynchronize(stream_));
            poll_reports();
            report_expect_[slot] = 0;
        }
        last_report_slot_ = slot;
        report_tag_ = report_tag_ + 1u ? report_tag_ + 1u : 1u;
        report_expect_[slot] = report_tag_;
        fb.report = d_reports_ + slot; fb.report_tag = report_tag_;
    }
    last_noalloc_ = noalloc; last_counters_ = bp.counters; last_vis_ = bp.vis; pending_want_hdr_ = want_hdr;
    last_frame_out_ = noalloc && world_ <= 1 ? noalloc_out : d_frame_.ptr;
    last_tiles_out_ = fb.tiles_rgba8;
    fb.packed = want_packed_ ? (ext_packed_ && !noalloc ? ext_packed_ : d_packed_.ptr) : nullptr;
    last_packed_out_ = fb.packed;
    fb.hdr = want_hdr ? d_hdr_.ptr : nullptr;

    // per-kernel timing events on a sample of the frames (every kTimingStride-th, and the first few):
    // eight event records per frame are a noticeable share of the host's enqueue time
    const bool timed = timed_frame;
    if (force_timed_) --force_timed_;
    ++frame_counter_;
    hipEvent_t* ev_ = nullptr;
    EventSet* timed_set = nullptr;
    if (timed) {
        EventSet& es = event_sets_[event_next_];
        if (es.pending) harvest(es);
        last_set_ = (int)event_next_;
        event_next_ = (event_next_ + 1) % kEventSets;
        ev_ = es.ev;
        timed_set = &es;
    }
    auto mark = [&](int k, hipStream_t on) { if (ev_) HIP_CHECK(hipEventRecord(ev_[k], on)); };
    auto sub_prev = now();
    auto sublap = [&](int k) { if (prof) { auto t = now(); sub[k] += std::chrono::duration<double, std::micro>(t - sub_prev).count(); sub_prev = t; } };
    auto emit = [&]() {
    bool tables_pending = true;
    sublap(5);
    auto clear = [&](const FrameBuffers& b, const FrameConst& c, hipStream_t on) {
        if (tables_pending) launch_clear(b, c, on, st.host_dev, tb, total);
        else launch_clear(b, c, on);
        tables_pending = false;
    };
    // shadow passes beside the main pass's geometry (their own stream and buffer set) unless the frame is timed
    const bool shadows_beside = shadows_beside_frame && !timed;
    hipStream_t const ss = shadows_beside ? shadow_stream_ : gs;
    // shadow cascades: depth-only passes of the same pipeline (cull, transform, setup, bins, raster)
    // into kShadowSize^2 visibility buffers, one per (directional light, cascade)
    // the buffers only the geometry phase uses exist once: geometry phases follow each other
    // (a frame whose geometry runs on another stream than the previous frame's — the launch mode changed — waits for
    // everything that stream holds: one event recorded then, so that frames of the one-stream mode record none at all;
    // two event records a frame cost C1' 5 % of its frame rate)
    if (last_geo_stream_ && last_geo_stream_ != gs) {
        HIP_CHECK(hipEventRecord(switch_ev_, last_geo_stream_));
        HIP_CHECK(hipStreamWaitEvent(gs, switch_ev_, 0));
    }
    last_geo_stream_ = gs;
    if (overlap) {
        // this parity's buffers (shadow layers included) were last read by the shading of the frame before
        // the previous one; a frame with timing events runs alone so that its kernel durations are its own
        HIP_CHECK(hipStreamWaitEvent(gs, shade_done_[p], 0));
        if (timed || serialize_next_) HIP_CHECK(hipStreamWaitEvent(gs, shade_done_[p ^ 1u], 0));
    }
    sublap(0);
    mark(7, gs);
    if (shadows_beside) {
        // the main pass's clear goes first: it brings the frame's tables to the device, which the shadow passes read too
        counter_set_ran_[0] = 1;
        clear(fb, fc, gs);
        HIP_CHECK(hipEventRecord(tables_ready_, gs));
        HIP_CHECK(hipStreamWaitEvent(ss, tables_ready_, 0));     // (also orders the shadow layers of this parity behind the shading that last read them)
    }
    for (uint32_t k = 0; k < n_sets; ++k) {
        // a cascade no caster can reach stays cleared and out of the pass: its share of the launch-bound kernels (and a
        // 33 MB clear) was most of a small scene's frame (C1': three of the four cascades)
        if (!clear_mask[k]) continue;
        DeviceBuffer<TileRef>& order = d_shadow_order_[clear_mask[k]];
        const uint32_t n_order = (uint32_t)__builtin_popcount(clear_mask[k]) * kShadowTiles;
        if (!order.ptr) {
            std::vector<TileRef> list;
            list.reserve(n_order);
            for (uint32_t c = 0; c < kShadowCascades; ++c)
                if ((clear_mask[k] >> c) & 1u)
                    for (uint32_t t = 0; t < kShadowTiles; ++t) list.push_back(TileRef{c * kShadowTiles + t, c * kShadowTiles + t});
            order.ensure(n_order);
            HIP_CHECK(hipMemcpy(order.ptr, list.data(), n_order * sizeof(TileRef), hipMemcpyHostToDevice));
        }
        FrameBuffers sfb = fb;
        sfb.items = reinterpret_cast<const DrawItem*>(tb + off_si); sfb.lights = nullptr; sfb.shadow_sets = nullptr; sfb.alpha_test = 0;
        sfb.clip = nullptr;        // depth only: nothing reads clip positions (k_transform skips the store)
        sfb.early_z = 0u; sfb.setup_pass = 0u;      // (the shadow passes run the plain multi-view kernels: one list of units)
        if (shadows_beside) {
            GeoSet& g = shadow_geo_;
            sfb.snap = g.snap.ptr; sfb.records = g.records.ptr; sfb.tilewords = reinterpret_cast<uint2*>(g.tilewords.ptr); sfb.clip_queue = g.clip_queue.ptr;
            sfb.chunk_count = g.chunk_count.ptr; sfb.active_units = reinterpret_cast<uint2*>(g.active_units.ptr); sfb.vblock_needed = g.vblock_needed.ptr;
            sfb.entries = g.entries.ptr; sfb.wide_list = g.wide_list.ptr; sfb.tile_slices = g.tile_slices.ptr; sfb.raster_extra = g.raster_extra.ptr;
            sfb.tile_count = g.tile_count.ptr; sfb.tile_offset = g.tile_offset.ptr; sfb.tile_cursor = g.tile_cursor.ptr;
        }
        sfb.counters = bp.counters + 1 + k;
        sfb.stat_stripes = bp.stripes + (size_t)(1 + k) * kStatStripes * kStatStride;
        sfb.launch_order = order.ptr; sfb.n_owned = n_order;
        sfb.vis = shadow_layers_[p][k].ptr;         // the four layers, contiguous: the stacked target's tiles
        sfb.frame_rgba8 = nullptr; sfb.tiles_rgba8 = nullptr; sfb.packed = nullptr; sfb.hdr = nullptr;
        const FrameConst& sfc = consts[1 + k];
        sfb.frame_const = const_slot(1 + k);
        clear(sfb, sfc, ss);
        counter_set_ran_[1 + k] = 1;
        for (uint32_t c = 0; c < kShadowCascades; ++c)
            if ((clear_mask[k] >> c) & 1u) shadow_layer_clear_[p][k][c] = !((draw_mask[k] >> c) & 1u);
        if (!draw_mask[k]) continue;
        launch_cull(sfb, sfc, ss);
        launch_transform(sfb, sfc, ss);
        launch_setup(sfb, sfc, cap_, ss);
        if (frame_no_chain_) continue;
        launch_clip(sfb, sfc, cap_, ss);
        launch_scan(sfb, sfc, cap_, ss);
        launch_bin(sfb, sfc, cap_, ss);
        launch_raster(sfb, sfc, cap_, ss);
    }
    if (shadows_beside) HIP_CHECK(hipEventRecord(shadows_done_, ss));
    mark(0, gs);
    counter_set_ran_[0] = 1;
    if (!shadows_beside) clear(fb, fc, gs);       // visibility words, tile counts, vertex-block flags, frame counters
    mark(9, gs);
    launch_cull(fb, fc, gs);       // work units that can reach the view volume and this context's stripes
    mark(10, gs);
    launch_transform(fb, fc, gs);
    mark(1, gs);
    launch_setup(fb, fc, cap_, gs);
    mark(11, gs);
    if (!frame_no_chain_) launch_clip(fb, fc, cap_, gs);
    mark(2, gs);
    if (!frame_no_chain_) launch_scan(fb, fc, cap_, gs);
    mark(3, gs);
    if (!frame_no_chain_) launch_bin(fb, fc, cap_, gs);
    mark(4, gs);
    if (!frame_no_chain_) launch_raster(fb, fc, cap_, gs);
    mark(5, gs);
    sublap(1);
    if (overlap) HIP_CHECK(hipEventRecord(geo_done_[p], gs));
    if (overlap) HIP_CHECK(hipStreamWaitEvent(stream_, geo_done_[p], 0));
    if (shadows_beside) HIP_CHECK(hipStreamWaitEvent(stream_, shadows_done_, 0));
    sublap(2);
    mark(8, stream_);       // shading starts here on its own stream (the hand-over from the geometry stream takes a few us)
    {
        uint64_t ran = 0;
        for (uint32_t k = 0; k < n_counter_sets_ && k < 64u; ++k) if (counter_set_ran_[k]) ran |= 1ull << k;
        fb.ran_mask = ran;
    }
    launch_shade(fb, fc, stream_);
    const OverflowGate gate{fb.all_counters, fb.n_counter_sets, fb.no_chain, fb.ran_mask};
    if (scaled())   // show_texture.wgsl over the surface (visual_server.rs:629-633: filtered sampler above factor 1)
        launch_present(d_hdr_.ptr, width_, height_, d_present_.ptr, surface_w_, surface_h_, render_size_factor_ > 1.0f, tone_mapping_, gate, stream_);
    if (overlay) {
        // Pipeline2d::render + the second draw of Backend::render, over the frame just presented
        OverlayParams op{};
        op.prims = reinterpret_cast<const Prim2D*>(tb + off_pr); op.n_prims = (uint32_t)prims.size();
        op.srgb_lut = d_srgb_lut_.ptr;
        op.font = TextureDesc{d_font_.ptr, font_w_, font_h_, 1u, 0u};
        op.fullscreen = (uint32_t)fullscreen_;
        if (fullscreen_ == 1) { op.depth_src = bp.vis; op.src_w = width_; op.src_h = height_; op.src_tiles_x = tiles_x_; }
        else if (fullscreen_ == 2) {
            for (uint32_t k = 0; k < n_sets; ++k)
                if (lights_[shadowed[k]].id == fullscreen_light_) {
                    op.depth_src = shadow_layers_[p][k].ptr;       // layer 0 (build_fullscreen_texture_array_bind_group(.., 0))
                    op.src_w = op.src_h = kShadowSize; op.src_tiles_x = kShadowSize / kTile;
                }
        }
        op.sw = surface_w_; op.sh = surface_h_; op.tiles_x = (surface_w_ + kTile - 1) / kTile;
        op.rank = rank_; op.world = world_;
        op.frame = world_ > 1 ? nullptr : reinterpret_cast<uint32_t*>(scaled() ? d_present_.ptr : d_frame_.ptr);
        op.tiles = world_ > 1 ? reinterpret_cast<uint32_t*>(fb.tiles_rgba8) : nullptr;
        op.overlay_out = d_overlay_.ptr;
        op.gate = gate;
        if (world_ > 1) HIP_CHECK(hipMemsetAsync(d_overlay_.ptr, 0, (size_t)surface_w_ * surface_h_ * 4, stream_));
        launch_overlay(op, op.tiles_x * ((surface_h_ + kTile - 1) / kTile), stream_);
    }
    mark(6, stream_);
    sublap(3);
    if (overlap) HIP_CHECK(hipEventRecord(shade_done_[p], stream_));
    if (tune_mark_ >= 0) { HIP_CHECK(hipEventRecord(tune_ev_[tune_mark_], stream_)); tune_mark_ = -1; }
    sublap(4);
    };
    emit();
    lap(3);
    if (timed_set) timed_set->pending = true;
    if (noalloc) HIP_CHECK(hipEventRecord(g_scratch_busy[device_], stream_));
    // the pinned block is free again once this frame's first kernel has read it; recorded after the
    // whole geometry sequence (three blocks rotate)
    HIP_CHECK(hipEventRecord(st.done, gs));
    st.in_flight = true;
    serialize_next_ = timed;
    // the frame's counters come back in finish(): only a frame somebody waits for needs them
    HIP_CHECK(hipGetLastError());
    lap(4);
    if (prof && ++acc_n == 256) {
        std::fprintf(stderr, "rnd host us/frame: build+ensure %.1f, staging wait %.1f, stage+h2d %.1f, launches %.1f, d2h %.1f\n",
                     acc[0] / 256, acc[1] / 256, acc[2] / 256, acc[3] / 256, acc[4] / 256);
        std::fprintf(stderr, "  launches: stream waits %.1f, geometry launches %.1f, hand-over events %.1f, shading launch %.1f, shade_done record %.1f\n",
                     sub[0] / 256, sub[1] / 256, sub[2] / 256, sub[3] / 256, sub[4] / 256);
        for (double& a : acc) a = 0; acc_n = 0;
        for (double& a : sub) a = 0;
    }
    frame_pending_ = true;
    stats_ = rnd_frame_stats{};
    stats_.n_vertices = fc.n_verts; stats_.n_triangles = fc.n_tris; stats_.n_draw_items = fc.n_items;
    stats_.n_lights = fc.n_lights; stats_.n_tiles = n_owned_;
    stats_.early_depth = early_z_frame_ && !frame_alpha_test_ && msaa_ == 1 ? 1u : 0u;
    stats_.n_pixels = 0;
}

// reference: VisualServer::render (visual_server.rs:224-311).  The notice about an earlier truncated frame is
// raised AFTER this call's frame has been enqueued (see rnd.h).
void VisualServer::render() {
    enqueue_frame(false);
    if (truncated_notice_) {
        truncated_notice_ = false;
        throw Error(RND_ERR_TRUNCATED, "an earlier frame overflowed an internal stream and was not presented; the streams have been grown and this frame is enqueued");
    }
}

void VisualServer::note_chain(bool needed, bool failed) {
    if (failed) { full_chain_frames_ = kFullChainFrames; chain_free_streak_ = 0; }
    else if (needed) chain_free_streak_ = 0;
    else if (chain_free_streak_ < 0xFFFFFFFFu) ++chain_free_streak_;
}

void VisualServer::note_depth_complexity(uint64_t n_fragments) {
    // (n_fragments: pixel centres in the bboxes of the small triangles, about twice what they cover: C3 1.4 per pixel,
    // C4 — ten layers — 30)
    const double per_sample = (double)n_fragments / ((double)width_ * height_ * msaa_);
    // (while the plain variant runs, the count is exact and gives the rate at which the early-depth variant's second pass
    // counts the triangles of the units it turns away whole: bbox pixels per input triangle, x 16)
    if (!early_z_state_ && fc_.n_tris) bbox_px_hint_x16_ = (uint32_t)std::min<uint64_t>(n_fragments * 16u / fc_.n_tris, 0xFFFFu);
    if (per_sample >= 6.0) early_z_state_ = true;
    else if (per_sample < 4.0) early_z_state_ = false;
}

void VisualServer::grow_streams(uint32_t overflow, uint32_t n_records, uint32_t n_clip, uint32_t n_entries) {
    if (overflow & 1u) clip_records_ = std::max<uint32_t>(clip_records_, (uint32_t)std::min<uint64_t>((uint64_t)n_records + n_records / 4 + 4096, 0x7FFFFFF0ull));
    if (overflow & 2u) cap_.clip_queue = std::max<uint32_t>(cap_.clip_queue, (uint32_t)std::min<uint64_t>((uint64_t)n_clip + n_clip / 4 + 4096, 0xFFFFFFF0ull));
    if (overflow & 4u) cap_.entries = std::max<uint32_t>(cap_.entries, (uint32_t)std::min<uint64_t>((uint64_t)n_entries + n_entries / 4 + 4096, 0xFFFFFFF0ull));
}

// Looks at the reports of the frames in flight without waiting for any: a report whose tag has arrived belongs to
// a frame whose shading kernel has started; if its geometry overflowed, the streams grow to what it asked for.
void VisualServer::poll_reports() {
    for (uint32_t s = 0; s < kReportSlots; ++s) {
        if (!report_expect_[s]) continue;
        const volatile FrameReport* r = &h_reports_[s];
        if (r->frame_tag != report_expect_[s]) continue;       // not there yet
        report_expect_[s] = 0;
        note_depth_complexity(r->n_fragments);
        const uint32_t flags = r->overflow, ovf = flags & 7u;
        note_chain((flags & kReportChainNeeded) != 0u || ovf != 0u, (flags & kReportNoChainFailed) != 0u);
        if (flags & kReportNoChainFailed) truncated_notice_ = true;        // (withheld: the frame is lost unless rnd_finish re-runs it)
        if (!ovf) continue;
        grow_streams(ovf, r->n_records, r->n_clip, r->n_entries);
        truncated_notice_ = true;
    }
}

// rnd_render_noalloc: the same frame with every intermediate stream in the static scratch array and the image in
// the caller's buffer.  No hipMalloc/hipFree/hipHostMalloc on this path (pinned blocks and events date from the
// constructor; the host tables reuse their capacity after the first frames).  One stream, no overlap with the
// previous frame: every context on this GPU shares the array, so calls that overlap on one GPU race.
void VisualServer::render_noalloc(void* out_dev, uint64_t out_bytes) {
    if (!out_dev) throw Error(RND_ERR_INVALID, "rnd_render_noalloc: null output buffer");
    enqueue_frame(false, static_cast<uint8_t*>(out_dev), out_bytes);
}

void VisualServer::finish() {
    if (tune_state_ == kTuneProbeOverlap || tune_state_ == kTuneProbeSerial) { tune_state_ = kTuneIdle; tune_mark_ = -1; }
    if (!frame_pending_) return;
    use_device();
    uint32_t grew = 0;
    for (int attempt = 0; attempt < 8; ++attempt) {
        HIP_CHECK(hipMemcpyAsync(h_counters_, last_counters_, n_counter_sets_ * sizeof(FrameCounters), hipMemcpyDeviceToHost, stream_));
        HIP_CHECK(hipStreamSynchronize(stream_));
        frame_pending_ = false;
        FrameCounters c = h_counters_[0];
        for (uint32_t k = 1; k < n_counter_sets_; ++k) {       // shadow passes share the streams: take the largest demand
            if (!counter_set_ran_[k]) continue;                 // a skipped cascade: nothing wrote its slot this frame
            const FrameCounters& o = h_counters_[k];
            c.overflow |= o.overflow;
            c.n_records = std::max(c.n_records, o.n_records); c.n_clip = std::max(c.n_clip, o.n_clip);
            c.n_entries = std::max(c.n_entries, o.n_entries);
        }
        bool chain_needed = c.chain_needed != 0u;
        for (uint32_t k = 1; k < n_counter_sets_; ++k) if (counter_set_ran_[k]) chain_needed = chain_needed || h_counters_[k].chain_needed != 0u;
        if (c.overflow & ~7u) {
            char msg[96];
            std::snprintf(msg, sizeof msg, "internal consistency check failed in a kernel (flags 0x%x)", c.overflow);
            throw Error(RND_ERR_DEVICE, msg);
        }
        if (!c.overflow && frame_no_chain_ && chain_needed) {
            // launched without the record chain, needed it, was withheld: run it again in full (and its report is handled here)
            if (last_report_slot_ < kReportSlots) report_expect_[last_report_slot_] = 0;
            note_chain(true, true);
            if (attempt == 7) throw Error(RND_ERR_DEVICE, "frame could not be completed");
            enqueue_frame(pending_want_hdr_);
            continue;
        }
        if (!c.overflow) break;
        // an internal stream was too small: grow it to what the frame asked for and run it again
        grow_streams(c.overflow, c.n_records, c.n_clip, c.n_entries);
        grew = 1;
        note_chain(true, frame_no_chain_ && chain_needed);
        // (this frame is re-run here and delivered: its own report must not come back as the notice of a dropped frame)
        if (last_report_slot_ < kReportSlots) report_expect_[last_report_slot_] = 0;
        if (last_noalloc_) {
            // growing means allocating: the frame in the caller's buffer is incomplete and stays so; the next
            // frame (either entry point) is laid out with the capacities learnt here
            have_frame_ = false;
            throw Error(RND_ERR_OOM, "rnd_render_noalloc: the frame overflowed an internal stream (records, clip queue or bin entries); "
                                     "render it again, or once through rnd_render");
        }
        if (attempt == 7) throw Error(RND_ERR_OOM, "frame does not fit the internal streams");
        enqueue_frame(pending_want_hdr_);
    }
    const FrameCounters c = *h_counters_;
    if (std::getenv("RND_DEBUG_COUNTERS")) std::fprintf(stderr, "rnd counters: units %u active %u near %u far %u valid %u frags %u\n", fc_.n_units, c.n_active, c.n_near, c.n_far, c.n_valid, c.n_fragments);
    stats_.n_setup = c.n_valid; stats_.n_bin_entries = c.n_entries; stats_.overflow = grew;
    stats_.n_fragments = c.n_fragments;
    stats_.n_units_hidden = c.n_far - std::min(c.n_far, c.n_active - std::min(c.n_active, c.n_near));      // (both zero outside a two-pass early-depth frame)
    note_depth_complexity(c.n_fragments);
    // pixels this context rasterizes: owned tiles clipped to the frame
    uint64_t px = 0;
    for (uint32_t t : owned_tiles(rank_, world_)) {
        uint32_t tx = t % tiles_x_, ty = t / tiles_x_;
        px += (uint64_t)std::min(kTile, width_ - tx * kTile) * std::min(kTile, height_ - ty * kTile);
    }
    stats_.n_pixels = px;
    for (auto& es : event_sets_)
        if (es.pending) harvest(es);
    have_frame_ = true;
}

// Reads one completed frame's events into timings_ and the running sums.
void VisualServer::harvest(EventSet& es) {
    HIP_CHECK(hipEventSynchronize(es.ev[6]));
    auto span = [&](int from, int to) { float ms = 0.0f; HIP_CHECK(hipEventElapsedTime(&ms, es.ev[from], es.ev[to])); return ms; };
    timings_ = rnd_timings{};
    timings_.clear_ms = span(0, 9); timings_.cull_ms = span(9, 10); timings_.transform_ms = span(10, 1);
    timings_.setup_ms = span(1, 11); timings_.clip_ms = span(11, 2); timings_.scan_ms = span(2, 3);
    timings_.bin_ms = span(3, 4); timings_.raster_ms = span(4, 5); timings_.shade_ms = span(8, 6);
    timings_.shadow_ms = span(7, 0); timings_.frame_ms = span(7, 6);
    const float* f = reinterpret_cast<const float*>(&timings_);
    for (int k = 0; k < kTimingFields; ++k) { timing_sum_[k] += f[k]; timing_max_[k] = std::max(timing_max_[k], f[k]); }
    timing_frames_ += 1;
    have_timings_ = true;
    es.pending = false;
}

rnd_timings VisualServer::mean_timings(uint32_t& n_frames, bool reset, rnd_timings* max_out) {
    finish();
    rnd_timings m{}, mx{};
    n_frames = timing_frames_;
    if (timing_frames_) {
        const double inv = 1.0 / timing_frames_;
        float* f = reinterpret_cast<float*>(&m);
        float* g = reinterpret_cast<float*>(&mx);
        for (int k = 0; k < kTimingFields; ++k) { f[k] = (float)(timing_sum_[k] * inv); g[k] = timing_max_[k]; }
    }
    if (max_out) *max_out = mx;
    if (reset) { for (double& v : timing_sum_) v = 0.0; for (float& v : timing_max_) v = 0.0f; timing_frames_ = 0; }
    return m;
}

void VisualServer::set_tiles_output(void* dev_ptr, uint64_t bytes) {
    // only read when the next frame is enqueued: frames already in flight keep the buffer they were
    // launched with, so a caller may alternate two buffers without draining the stream
    if (dev_ptr && bytes < (uint64_t)n_owned_ * kTilePixels * 4) throw Error(RND_ERR_INVALID, "tile output buffer too small");
    ext_tiles_ = static_cast<uint8_t*>(dev_ptr);
    ext_tiles_bytes_ = bytes;
}

void VisualServer::need_frame() const {
    if (!have_frame_) throw Error(RND_ERR_NO_FRAME, "no frame rendered yet");
}

// ---- read-backs -----------------------------------------------------------------------------
void VisualServer::read_color_rgba8(uint8_t* dst) {
    if (!dst) throw Error(RND_ERR_INVALID, "null destination");
    finish(); need_frame(); use_device();
    if (assembled_frame_) {     // rank 0 after rnd_gather_tiles / rnd_composite_packed: the whole frame, as assembled
        HIP_CHECK(hipMemcpyAsync(dst, d_frame_.ptr, (size_t)width_ * height_ * 4, hipMemcpyDeviceToHost, stream_));
        HIP_CHECK(hipStreamSynchronize(stream_));
        return;
    }
    if (world_ > 1) {   // a partitioned context holds only its tiles: assemble them over a cleared frame
        HIP_CHECK(hipMemsetAsync(d_frame_.ptr, 0, (size_t)width_ * height_ * 4, stream_));
        detile(rank_, world_, last_tiles_out_);
    }
    if (scaled()) HIP_CHECK(hipMemcpyAsync(dst, d_present_.ptr, (size_t)surface_w_ * surface_h_ * 4, hipMemcpyDeviceToHost, stream_));
    else HIP_CHECK(hipMemcpyAsync(dst, world_ > 1 ? d_frame_.ptr : last_frame_out_, (size_t)width_ * height_ * 4, hipMemcpyDeviceToHost, stream_));
    HIP_CHECK(hipStreamSynchronize(stream_));
}

void VisualServer::read_color_hdr(float* dst) {
    if (!dst) throw Error(RND_ERR_INVALID, "null destination");
    finish(); need_frame(); use_device();
    // the frame path never stores HDR; run the frame once more with the HDR plane attached
    const size_t n = (size_t)width_ * height_ * 3;
    d_hdr_.ensure(n);
    enqueue_frame(true);
    finish();
    if (world_ > 1) {
        // pixels outside the owned tiles are not written
    }
    HIP_CHECK(hipMemcpyAsync(dst, d_hdr_.ptr, n * 4, hipMemcpyDeviceToHost, stream_));
    HIP_CHECK(hipStreamSynchronize(stream_));
}

void VisualServer::read_planes(float* depth, uint32_t* ids, uint8_t* coverage) {
    finish(); need_frame(); use_device();
    const size_t n = (size_t)width_ * height_;
    DeviceBuffer<float> d_depth; DeviceBuffer<uint32_t> d_ids; DeviceBuffer<uint8_t> d_cov;
    try {
        if (depth) { d_depth.ensure(n); HIP_CHECK(hipMemsetAsync(d_depth.ptr, 0, n * 4, stream_)); }
        if (ids) { d_ids.ensure(n); HIP_CHECK(hipMemsetAsync(d_ids.ptr, 0xFF, n * 4, stream_)); }
        if (coverage) { d_cov.ensure(n); HIP_CHECK(hipMemsetAsync(d_cov.ptr, 0, n, stream_)); }
        launch_extract(last_vis_, d_launch_order_.ptr, n_owned_, fc_, d_depth.ptr, d_ids.ptr, d_cov.ptr, stream_);
        if (depth) HIP_CHECK(hipMemcpyAsync(depth, d_depth.ptr, n * 4, hipMemcpyDeviceToHost, stream_));
        if (ids) HIP_CHECK(hipMemcpyAsync(ids, d_ids.ptr, n * 4, hipMemcpyDeviceToHost, stream_));
        if (coverage) HIP_CHECK(hipMemcpyAsync(coverage, d_cov.ptr, n, hipMemcpyDeviceToHost, stream_));
        HIP_CHECK(hipStreamSynchronize(stream_));
    } catch (...) {
        d_depth.release(); d_ids.release(); d_cov.release();
        throw;
    }
    d_depth.release(); d_ids.release(); d_cov.release();
}

rnd_timings VisualServer::last_timings() { finish(); need_frame(); return timings_; }
rnd_frame_stats VisualServer::frame_stats() { finish(); need_frame(); return stats_; }

// ---- multi-GPU ---------------------------------------------------------------------------------
void VisualServer::tiles_device_ptr(void** p, uint64_t* bytes) {
    if (world_ <= 1) throw Error(RND_ERR_INVALID, "no tile partition set");
    need_frame();
    *p = ext_tiles_ ? ext_tiles_ : d_tiles_.ptr; *bytes = (uint64_t)n_owned_ * kTilePixels * 4;
}

void VisualServer::detile(uint32_t src_rank, uint32_t world, const void* dev_tiles) {
    if (!dev_tiles || world == 0 || src_rank >= world) throw Error(RND_ERR_INVALID, "bad detile arguments");
    use_device();
    // tile lists per (rank, world) are built once and stay on the device; the launch is asynchronous
    const uint64_t key = ((uint64_t)world << 32) | src_rank;
    auto it = detile_lists_.find(key);
    if (it == detile_lists_.end() || it->second.tiles_x != tiles_x_ || it->second.tiles_y != tiles_y_) {
        if (it != detile_lists_.end()) { it->second.list.release(); detile_lists_.erase(it); }
        std::vector<uint32_t> list = owned_tiles(src_rank, world);
        DetileList dl;
        dl.tiles_x = tiles_x_; dl.tiles_y = tiles_y_; dl.n = (uint32_t)list.size();
        dl.list.ensure(std::max<size_t>(list.size(), 1));
        if (!list.empty()) HIP_CHECK(hipMemcpy(dl.list.ptr, list.data(), list.size() * 4, hipMemcpyHostToDevice));
        it = detile_lists_.emplace(key, dl).first;
    }
    if (!it->second.n) return;
    FrameConst fc = fc_;
    fc.width = width_; fc.height = height_; fc.tiles_x = tiles_x_; fc.tiles_y = tiles_y_;
    launch_detile(static_cast<const uint8_t*>(dev_tiles), it->second.list.ptr, it->second.n, fc, d_frame_.ptr, stream_);
    HIP_CHECK(hipGetLastError());
}

void VisualServer::detile_gathered(uint32_t world, const void* dev_gathered, uint64_t stride_bytes) {
    if (!dev_gathered || world == 0 || stride_bytes % 4) throw Error(RND_ERR_INVALID, "bad detile arguments");
    // the largest share is the first slot's (the last rank's: stripe_owner)
    uint64_t rows0 = 0;
    for (uint32_t first = 0; first < tiles_y_; first += kOwnerGroup * world) rows0 += std::min(kOwnerGroup, tiles_y_ - first);
    if (stride_bytes < rows0 * tiles_x_ * kTilePixels * 4) throw Error(RND_ERR_INVALID, "gathered stride smaller than the largest rank's tiles");
    use_device();
    FrameConst fc = fc_;
    fc.width = width_; fc.height = height_; fc.tiles_x = tiles_x_; fc.tiles_y = tiles_y_;
    launch_detile_gathered(static_cast<const uint8_t*>(dev_gathered), stride_bytes, world, fc, d_frame_.ptr, stream_);
    HIP_CHECK(hipGetLastError());
    assembled_frame_ = true; have_frame_ = true;       // read-backs hand out the assembled frame until the next render
}

void VisualServer::packed_device_ptr(void** p, uint64_t* bytes) {
    use_device();
    if (scaled()) throw Error(RND_ERR_UNSUPPORTED, "the packed (sort-last) frame is not built for render_size_factor != 1");
    if (!want_packed_ || !have_frame_) {
        want_packed_ = true;
        finish();
        enqueue_frame(false);
        finish();
    }
    *p = const_cast<uint64_t*>(last_packed_out_ ? last_packed_out_ : d_packed_.ptr); *bytes = (uint64_t)width_ * height_ * 8;
}

void VisualServer::unpack(const void* dev_packed) {
    if (!dev_packed) throw Error(RND_ERR_INVALID, "null packed frame");
    use_device();
    FrameConst fc = fc_;
    fc.width = width_; fc.height = height_;
    launch_unpack(static_cast<const uint64_t*>(dev_packed), fc, d_frame_.ptr, nullptr, stream_);
    HIP_CHECK(hipGetLastError());      // (stream-ordered: read-backs and rnd_finish wait on the same stream)
    have_frame_ = true;
    last_frame_out_ = d_frame_.ptr;
    assembled_frame_ = true;
}

void VisualServer::set_packed_output(void* dev_ptr, uint64_t bytes) {
    // as rnd_set_tiles_output: read when the next frame is enqueued, so a caller may alternate two buffers
    if (dev_ptr && (bytes < (uint64_t)width_ * height_ * 8 || (reinterpret_cast<uintptr_t>(dev_ptr) & 15u)))
        throw Error(RND_ERR_INVALID, "packed output buffer too small or not 16-byte aligned");
    if (scaled()) throw Error(RND_ERR_UNSUPPORTED, "the packed (sort-last) frame is not built for render_size_factor != 1");
    ext_packed_ = static_cast<uint64_t*>(dev_ptr);
    if (dev_ptr) want_packed_ = true;
}

void VisualServer::min_packed(const void* dev_strips, uint32_t n, uint64_t stride_words, void* dev_out, uint64_t count) {
    if (!dev_strips || !dev_out || n == 0 || (stride_words & 1u) || stride_words < count ||
        ((reinterpret_cast<uintptr_t>(dev_strips) | reinterpret_cast<uintptr_t>(dev_out)) & 15u))
        throw Error(RND_ERR_INVALID, "bad strips: null, odd stride, stride below count, or not 16-byte aligned");
    use_device();
    launch_min_packed(static_cast<const uint64_t*>(dev_strips), n, stride_words, static_cast<uint64_t*>(dev_out), count, stream_);
    HIP_CHECK(hipGetLastError());
}

void VisualServer::set_stream(hipStream_t s) {
    use_device();
    finish();
    HIP_CHECK(hipStreamSynchronize(stream_));
    if (geo_stream_) HIP_CHECK(hipStreamSynchronize(geo_stream_));
    last_geo_stream_ = nullptr;          // (nothing is in flight: the next frame has nothing to wait for)
    if (own_stream_ && stream_) HIP_CHECK(hipStreamDestroy(stream_));
    if (s) { stream_ = s; own_stream_ = false; }
    else { HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking)); own_stream_ = true; }
}

void VisualServer::color_device_ptr(void** p, uint64_t* bytes) {
    if (scaled()) { *p = d_present_.ptr; *bytes = (uint64_t)surface_w_ * surface_h_ * 4; return; }
    *p = d_frame_.ptr; *bytes = (uint64_t)width_ * height_ * 4;
}

template struct DeviceBuffer<float>;
template struct DeviceBuffer<uint32_t>;
template struct DeviceBuffer<uint8_t>;
template struct DeviceBuffer<uint64_t>;

}  // namespace rnd
