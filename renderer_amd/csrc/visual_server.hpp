// Host side of the frame hot path: a C++ restatement of the reference's `VisualServer`
// (src/renderer/visual_server.rs:30-874) with the wgpu plumbing replaced by HIP allocations and
// kernel launches.  Method names and argument meaning follow the reference; the C ABI in capi.cpp
// is a 1:1 wrapper.
#pragma once
#include <hip/hip_runtime.h>

#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "device_types.h"
#include "kernels.h"
#include "rnd_math.hpp"

namespace rnd {

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string& m) : std::runtime_error(m), status(s) {}
};

void check_hip(hipError_t e, const char* what);
#define HIP_CHECK_COMM(x) ::rnd::check_hip((x), #x)
void comm_make_id(uint8_t* id);      // ncclGetUniqueId (comm.cpp)

using UniqueNodeId = uint64_t;   // src/scene.rs:14-15 flattened to 64 bits

struct Camera {                  // src/camera.rs:4-9
    float vfov = 1.3f, aspect_ratio = 1.667f, near = 0.05f, far = 100.0f;
};
struct Light {                   // src/light.rs:3-7, kind id as LightKind::id()
    float color[4] = {1, 1, 1, 1};
    uint32_t kind = RND_LIGHT_POINT;
    float radius = 1.0f;
};
struct Material {                // src/material.rs:3-8
    float base_color[4] = {1, 1, 1, 1};
    uint32_t billboard_mode = RND_BILLBOARD_OFF;
    bool unlit = false;
    uint32_t texture = RND_NONE;
};

template <class T>
struct DeviceBuffer {
    T* ptr = nullptr;
    size_t count = 0;
    void ensure(size_t n, bool keep = false);
    void release();
};

struct RenderSubmesh {           // visual_server.rs:948-953
    DeviceBuffer<rnd_vertex> vertex_buffer;
    DeviceBuffer<float> positions;       // px | py | pz, n_verts each
    DeviceBuffer<uint32_t> index_buffer;
    DeviceBuffer<ChunkInfo> chunks;      // one per kChunkTrisSmall triangles, then one per kChunkTris triangles
    uint32_t n_chunks = 0, n_chunks_small = 0;
    uint32_t index_count = 0, vertex_count = 0;
    uint32_t material = RND_NONE;
    float box_lo[3] = {0, 0, 0}, box_hi[3] = {0, 0, 0};   // object-space bounds (infinite on an axis with a non-finite coordinate)
};
struct RenderMesh { std::vector<RenderSubmesh> submeshes; };
struct RenderTexture {           // RenderScene::textures (visual_server.rs:901): RGBA8 levels, decoded at the tap
    DeviceBuffer<uint8_t> texels;        // the caller's RGBA8 mip chain, 4 B per texel
    uint32_t width = 0, height = 0, mips = 0;
    uint32_t slot = 0;           // index into the device texture table
    bool opaque = true;          // every alpha == 255
};
struct RenderLight {
    UniqueNodeId id; Mat4 transform; Light light;
    Mat4 cascades[4];     // RenderShadowCascade::projview x4 (visual_server.rs:380-405), as of the last set_light
};
struct RenderMeshInstance {      // visual_server.rs:955-962
    UniqueNodeId id;
    Mat4 transform;
    uint32_t mesh;
    uint32_t material_override;
    bool casts_shadows;
};

class VisualServer {
public:
    VisualServer(uint32_t width, uint32_t height, int device);
    ~VisualServer();
    VisualServer(const VisualServer&) = delete;

    // settings — visual_server.rs:156-190
    void render_size(uint32_t& w, uint32_t& h) const { w = surface_w_; h = surface_h_; }
    void render_resolution(uint32_t& w, uint32_t& h) const { w = width_; h = height_; }
    void set_render_size(uint32_t w, uint32_t h);
    float render_size_factor() const { return render_size_factor_; }
    void set_render_size_factor(float f);
    uint32_t msaa_sample_count() const { return msaa_; }
    void set_msaa(uint32_t n);
    uint32_t tone_mapping() const { return tone_mapping_; }
    void set_tone_mapping(uint32_t m);

    // resources — visual_server.rs:664-769
    void register_mesh(uint32_t mesh_id, uint32_t n_submeshes, const rnd_submesh_desc* submeshes);
    void register_material(uint32_t material_id, const Material& m);
    void register_texture(uint32_t id, uint32_t w, uint32_t h, uint32_t mips, const uint8_t* data);

    // per-frame state — visual_server.rs:206-222, :350-474, :481-550, :579-581
    void set_camera(const Mat4& transform, const Camera& camera);
    void set_light(UniqueNodeId id, const Mat4& transform, const Light& light);
    void set_mesh_instance(UniqueNodeId id, const Mat4& transform, uint32_t mesh_id, uint32_t material_override,
                           bool casts_shadows);
    void set_sprite(UniqueNodeId id, const Mat4& transform, uint32_t texture_id, const float color[4]);
    void reset_scene();

    // 2D overlay — visual_server.rs:192-204, :313-348, :476-479, :552-577 (Pipeline2d)
    void set_uiboxes(uint32_t n, const rnd_uibox* boxes);
    void set_font_image(uint32_t w, uint32_t h, const uint8_t* rgba8);
    void reset_texts();
    void add_text(const uint8_t* text, uint32_t n, float x, float y, float font_size);
    void set_depth_fullscreen_texture();
    void set_shadow_map_fullscreen_texture(UniqueNodeId light);
    void unset_fullscreen_texture();
    void read_overlay_rgba8(uint8_t* dst);

    // frame — visual_server.rs:224-311
    void render();
    void render_noalloc(void* out_dev, uint64_t out_bytes);
    void finish();
    static constexpr size_t kScratchBytes = 128ull << 20;      // the static device array behind render_noalloc

    // read-backs
    void read_color_rgba8(uint8_t* dst);
    void read_color_hdr(float* dst);
    void read_planes(float* depth, uint32_t* ids, uint8_t* coverage);
    rnd_timings last_timings();
    rnd_timings mean_timings(uint32_t& n_frames, bool reset, rnd_timings* max_out = nullptr);
    void time_frames(uint32_t n) { force_timed_ = n; }
    rnd_frame_stats frame_stats();

    // multi-GPU
    void set_tile_partition(uint32_t rank, uint32_t world);
    void tile_info(uint32_t& tile_px, uint32_t& n_owned, uint32_t& n_total) const;
    void tiles_device_ptr(void** p, uint64_t* bytes);
    void set_tiles_output(void* dev_ptr, uint64_t bytes);
    void detile(uint32_t src_rank, uint32_t world, const void* dev_tiles);
    void detile_gathered(uint32_t world, const void* dev_gathered, uint64_t stride_bytes);
    void packed_device_ptr(void** p, uint64_t* bytes);
    void unpack(const void* dev_packed);
    void set_stream(hipStream_t s);
    void color_device_ptr(void** p, uint64_t* bytes);
    void set_packed_output(void* dev_ptr, uint64_t bytes);
    void min_packed(const void* dev_strips, uint32_t n, uint64_t stride_words, void* dev_out, uint64_t count);
    // RCCL exchanges from C (comm.cpp)
    void comm_init(const uint8_t* id, uint32_t rank, uint32_t world);
    void comm_destroy();
    void gather_tiles();
    void composite_packed();

    // for tests of the host logic
    const Mat4& projection() const { return projection_; }
    const Mat4& view() const { return view_; }

private:
    void use_device() const;
    void rebuild_tiles();
    std::vector<uint32_t> owned_tiles(uint32_t rank, uint32_t world) const;
    void build_frame(std::vector<DrawItem>& items, std::vector<uint32_t>& vert_bases, std::vector<uint32_t>& tri_bases,
                     std::vector<uint32_t>& unit_bases,
                     std::vector<GpuLight>& lights, std::vector<DrawItem>& shadow_items);
    void enqueue_frame(bool want_hdr, uint8_t* noalloc_out = nullptr, uint64_t noalloc_bytes = 0);
    void need_frame() const;

    int device_;
    uint32_t width_, height_;            // the 3D render target: surface * render_size_factor, truncated
    uint32_t surface_w_ = 0, surface_h_ = 0;
    bool frame_alpha_test_ = false, frame_textured_ = false;      // set by build_frame
    void apply_size(uint32_t sw, uint32_t sh, float factor);
    bool scaled() const { return width_ != surface_w_ || height_ != surface_h_; }
    DeviceBuffer<uint8_t> d_present_;     // surface-sized RGBA8 when scaled()
    float render_size_factor_ = 1.0f;
    uint32_t msaa_ = 1;
    uint32_t tone_mapping_ = RND_TONE_MAPPING_REINHARD;
    uint32_t rank_ = 0, world_ = 1;
    bool want_packed_ = false;

    hipStream_t stream_ = nullptr;
    bool own_stream_ = true;
    // ring of per-frame event sets, so kernel timings can be harvested without a per-frame sync
    static constexpr int kEventSets = 16, kEventsPerSet = 12;     // [7] = before the shadow passes, [8] = before shading, [9..11] after clear / cull / setup
    struct EventSet {
        hipEvent_t ev[kEventsPerSet] = {};
        bool pending = false;
    };
    EventSet event_sets_[kEventSets];
    uint32_t event_next_ = 0;
    static constexpr uint64_t kTimingStride = 256;    // a timed frame runs un-overlapped (two serialization points): keep them rare
    uint64_t frame_counter_ = 0;
    int last_set_ = -1;
    static constexpr int kTimingFields = sizeof(rnd_timings) / sizeof(float);
    double timing_sum_[kTimingFields] = {};
    float timing_max_[kTimingFields] = {};
    uint32_t timing_frames_ = 0;
    uint32_t force_timed_ = 0;        // rnd_time_frames: this many further frames record timing events
    void harvest(EventSet& s);
    uint8_t* ext_tiles_ = nullptr;
    uint64_t ext_tiles_bytes_ = 0;
    uint64_t* ext_packed_ = nullptr;          // rnd_set_packed_output
    const uint64_t* last_packed_out_ = nullptr;
    void* comm_ = nullptr;                    // ncclComm_t
    uint32_t comm_rank_ = 0, comm_world_ = 0;
    DeviceBuffer<uint64_t> d_comm_a_, d_comm_b_;      // exchange buffers of gather_tiles / composite_packed

    // RenderScene — visual_server.rs:896-905 (insertion-ordered instead of HashMap order)
    std::unordered_map<uint32_t, RenderMesh> meshes_;
    std::unordered_map<uint32_t, Material> materials_;
    std::unordered_map<uint32_t, RenderTexture> textures_;
    DeviceBuffer<TextureDesc> d_textures_;
    void upload_texture_table();
    std::vector<RenderLight> lights_;
    std::vector<RenderMeshInstance> mesh_instances_;
    std::unordered_map<UniqueNodeId, uint32_t> mesh_instance_index_;     // id -> position in mesh_instances_ (draw order = insertion order)
    std::unordered_map<UniqueNodeId, uint32_t> sprite_materials_;
    uint32_t next_sprite_material_ = 0x80000000u;

    // 2D state: UiBoxInstance list, GlyphInstance list of every text, font atlas, debug view
    struct Glyph { float pos[2], scale[2]; uint32_t glyph; };       // pipeline2d/glyph_instance.rs:5-9
    std::vector<rnd_uibox> uiboxes_;
    std::vector<Glyph> glyphs_;
    DeviceBuffer<float> d_font_;          // RGBA f32 (Rgba8Unorm), one level
    uint32_t font_w_ = 0, font_h_ = 0;
    DeviceBuffer<float> d_srgb_lut_;
    DeviceBuffer<uint32_t> d_overlay_;    // the 2D target of the last frame that drew one
    int fullscreen_ = 0;                  // 0 none, 1 depth, 2 shadow map of fullscreen_light_
    UniqueNodeId fullscreen_light_ = 0;
    bool overlay_drawn_ = false;
    void build_prims(std::vector<Prim2D>& prims) const;

    // SceneUniform — visual_server.rs:887-894
    Mat4 projection_, view_, camera_transform_;
    Mat4 inv_projection_view_;
    float ambient_light_[4] = {0.3f, 0.5f, 0.9f, 0.04f};

    // device state
    // Consecutive frames overlap: the geometry phase of frame N+1 (clear .. raster, on geo_stream_) runs
    // beside the shading of frame N (on stream_).  What shading reads from the geometry phase exists twice
    // (index = frame parity): per-frame tables, clip positions (frames that keep them), visibility buffer, counters.
    hipStream_t geo_stream_ = nullptr;
    hipEvent_t geo_done_[2] = {}, shade_done_[2] = {};
    // The shadow passes of a frame and its main pass's geometry meet only in the shading kernel: with a second set of
    // the geometry-only buffers the shadow passes run on a stream of their own beside the main pass (RND_SHADOW_STREAM=0:
    // one after the other on the geometry stream, as in rounds 1 and 2; frames with timing events always do that).
    struct GeoSet {
        DeviceBuffer<SnapVert> snap; DeviceBuffer<SetupRecord> records;
        DeviceBuffer<uint32_t> tilewords, clip_queue, chunk_count, vblock_needed, active_units, tile_count, tile_offset, tile_cursor, entries, wide_list, tile_slices;
        DeviceBuffer<uint2> raster_extra;
        void release() {
            snap.release(); records.release(); tilewords.release(); clip_queue.release(); chunk_count.release(); vblock_needed.release(); active_units.release();
            tile_count.release(); tile_offset.release(); tile_cursor.release(); entries.release(); wide_list.release(); tile_slices.release(); raster_extra.release();
        }
    } shadow_geo_;
    hipStream_t shadow_stream_ = nullptr;
    hipEvent_t tables_ready_ = nullptr, shadows_done_ = nullptr;
    bool shadow_concurrent_ = true;
    hipEvent_t switch_ev_ = nullptr;     // recorded on the previous frame's geometry stream when the launch mode changes
    uint32_t parity_ = 0;              // buffers of the last enqueued frame
    bool overlap_ = true;              // RND_OVERLAP=0: everything on stream_
    // Which of the two launch modes is faster for small frames (shading below 60 us) is measured, not guessed: a window of
    // frames in each mode between two events on the context's stream, the better one kept for kTuneHold frames
    // (tune_overlap in visual_server.cpp; RND_TUNE=0 leaves the fixed rule of round 2: overlap unless shading < 30 us).
    enum TuneState { kTuneIdle, kTuneProbeOverlap, kTuneProbeSerial, kTuneWait, kTuneHolding };
    static constexpr uint32_t kTuneSettle = 8, kTuneWindow = 24, kTuneHold = 4096;
    bool tune_enabled_ = true, tune_decided_ = false, tune_choice_ = true;
    TuneState tune_state_ = kTuneIdle;
    uint32_t tune_i_ = 0;
    int tune_mark_ = -1;               // the event to record after this frame's shading, or -1
    hipEvent_t tune_ev_[4] = {};
    float tune_ms_[2] = {0.0f, 0.0f};  // last measured window: overlapped, serial
    bool tune_overlap(bool timed);
    void tune_reset() { tune_state_ = kTuneIdle; tune_decided_ = false; tune_mark_ = -1; }
    hipStream_t last_geo_stream_ = nullptr;   // where the previous frame's geometry ran (the two modes can alternate)
    bool have_timings_ = false;        // timings_ holds a measured frame
    bool serialize_next_ = false;      // the previous frame recorded timing events: keep this one's geometry off its shading
    DeviceBuffer<uint8_t> d_tables_[2];             // per-frame tables (items, bases, lights, shadow items and sets), one upload
    DeviceBuffer<float4> d_clip_[2];
    DeviceBuffer<SnapVert> d_snap_;
    DeviceBuffer<SetupRecord> d_records_;
    DeviceBuffer<uint32_t> d_tilewords_, d_clip_queue_, d_chunk_count_, d_vblock_needed_, d_active_units_, d_tile_count_, d_tile_offset_, d_tile_cursor_, d_entries_;
    DeviceBuffer<FrameCounters> d_counters_[2];     // [0] main pass, then one per shadow pass
    uint32_t n_counter_sets_ = 1, h_counter_sets_ = 0;
    // shadow cascades of the directional lights
    DeviceBuffer<TileRef> d_shadow_order_[16];      // per mask of cascades: the tiles of those layers in the stacked target
    std::vector<std::array<bool, 4>> shadow_layer_clear_[2];     // the layer holds nothing but the clear value (a cascade without casters is not rendered again)
    bool cascade_has_casters(const Mat4& cascade) const;
    std::vector<DeviceBuffer<uint64_t>> shadow_layers_[2];   // per frame parity and shadowed light: kShadowCascades visibility buffers
    DeviceBuffer<TileRef> d_launch_order_;
    DeviceBuffer<uint64_t> d_vis_[2];
    DeviceBuffer<uint8_t> d_frame_, d_tiles_;
    DeviceBuffer<uint64_t> d_packed_;
    DeviceBuffer<float> d_hdr_;
    struct DetileList {
        DeviceBuffer<uint32_t> list;
        uint32_t n = 0, tiles_x = 0, tiles_y = 0;
    };
    std::unordered_map<uint64_t, DetileList> detile_lists_;
    Capacities cap_{0, 0, 0, 0, kWideListCap};
    DeviceBuffer<uint32_t> d_wide_list_, d_tile_slices_;
    DeviceBuffer<uint2> d_raster_extra_;
    uint32_t clip_records_ = 0;   // room in the record stream for the clipper's sub-triangles

    // pinned staging ring for the per-frame tables
    struct Staging {
        void* host = nullptr;
        void* host_dev = nullptr;     // the same block as the device addresses it (hipHostGetDevicePointer)
        size_t bytes = 0;
        hipEvent_t done = nullptr;
        bool in_flight = false;
    };
    Staging staging_[3];
    uint32_t staging_next_ = 0;
    FrameCounters* h_counters_ = nullptr;   // pinned
    static constexpr uint32_t kPinnedCounterSets = 16;        // allocated at construction (one set per pass of a frame)
    static constexpr size_t kStagingBytes = 1u << 20;         // each staging block at construction (~4000 draw items)
    // no-allocation frame: this device's address of the static scratch array; what the last frame used
    uint8_t* scratch_ = nullptr;
    bool last_noalloc_ = false;
    const FrameCounters* last_counters_ = nullptr;
    const uint64_t* last_vis_ = nullptr;
    const uint8_t* last_frame_out_ = nullptr;
    std::vector<uint8_t> counter_set_ran_;    // per counter set of the last frame: its pass ran (a skipped cascade leaves its slot unwritten)
    bool pending_want_hdr_ = false;           // the frame finish() may have to run again wanted the HDR plane
    // Overflow reports of the frames in flight (FrameReport in mapped pinned memory, written by each frame's shading
    // kernel): looked at without synchronizing when the next frame is enqueued.
    static constexpr uint32_t kReportSlots = 64;          // frames in flight whose reports are awaited (a slot still awaited when its turn comes again is waited for)
    FrameReport* h_reports_ = nullptr;        // pinned, mapped
    FrameReport* d_reports_ = nullptr;        // the same block as the device addresses it
    uint32_t report_expect_[kReportSlots] = {};   // tag each slot is waiting for (0: none)
    uint32_t report_next_ = 0, report_tag_ = 0;
    uint32_t last_report_slot_ = kReportSlots;        // the slot of the last enqueued frame (kReportSlots: none)
    bool truncated_notice_ = false;           // a frame in flight was found truncated: the next render() says so once
    bool test_small_streams_ = false;         // RND_TEST_SMALL_STREAMS=1: start with tiny streams (overflow tests)
    void poll_reports();
    void grow_streams(uint32_t overflow, uint32_t n_records, uint32_t n_clip, uint32_t n_entries);
    const uint8_t* last_tiles_out_ = nullptr;
    // the per-frame host tables, kept between frames so that a steady state allocates nothing on the host either
    std::vector<DrawItem> items_, shadow_items_;
    std::vector<uint32_t> vert_bases_, tri_bases_, unit_bases_;
    std::vector<uint32_t> unit_order_;          // draw items in processing order (build_frame)
    std::vector<float> unit_order_key_;
    // Depth complexity of the frames seen (bbox pixels of the small triangles per sample, from the frame reports): at 6
    // and above the next frames run k_setup's early-depth variant with the items near to far, below 4 the plain one again.
    // RND_EARLY_DEPTH = 0 / 1 forces one (read per frame).
    bool early_z_state_ = false, early_z_frame_ = false;
    // Frames without the record chain.  A pass whose triangles all take k_setup's direct path needs none of k_clip,
    // k_scan_tiles, k_bin_fill, k_raster — four launches of a latency-bound chain that is most of a small frame.  After
    // kNoChainStreak reported frames in a row that needed none of them the next frames are launched without; a frame
    // that then does need them is withheld by k_shade (as an overflowed one: the image keeps the frame before, the
    // report says so, rnd_finish re-runs it) and the full chain comes back for kFullChainFrames frames.
    // RND_NO_CHAIN=0 never leaves the chain out.
    static constexpr uint32_t kNoChainStreak = 4, kFullChainFrames = 64;
    uint32_t chain_free_streak_ = 0, full_chain_frames_ = 0;
    bool frame_no_chain_ = false, allow_no_chain_ = true;
    void note_chain(bool needed, bool failed);
    bool assembled_frame_ = false;              // d_frame_ holds a frame assembled from every rank (gather_tiles / composite_packed), until the next render
    uint32_t split_units_ = 0;                  // units of the early-depth variant's first pass (build_frame)
    DeviceBuffer<uint64_t> d_stat_stripes_[2];  // k_setup's striped statistics, one block per counter set (by frame parity, as the counters)
    uint32_t bbox_px_hint_x16_ = 0;             // bbox pixels per input triangle (x 16) of the last plain frame reported (note_depth_complexity)
    DeviceBuffer<uint2> d_far_units_;           // early-depth frames: the far items' units between k_cull and k_cull_far
    DeviceBuffer<uint4> d_unit_bounds_;         // early-depth frames: k_cull's screen rect and nearest depth per active unit
    DeviceBuffer<float> d_hiz_;                 // farthest depth per 8x8 block between its two passes
    void note_depth_complexity(uint64_t n_fragments);
    std::vector<GpuLight> gpu_lights_;
    std::vector<FrameConst> consts_;

    uint32_t tiles_x_ = 0, tiles_y_ = 0, n_owned_ = 0;
    bool have_frame_ = false, frame_pending_ = false;
    FrameConst fc_{};
    rnd_frame_stats stats_{};
    rnd_timings timings_{};
};

}  // namespace rnd
