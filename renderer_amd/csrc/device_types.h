// POD types shared by the host frame driver and the HIP kernels.
#pragma once
#include <stdint.h>

#include "../../include/rnd.h"

namespace rnd {

constexpr uint32_t kTile = 32;                 // square screen tile edge in pixels
constexpr uint32_t kTilePixels = kTile * kTile;
constexpr uint32_t kDirectPixels = 32;        // K2a rasterizes a triangle itself when its bbox holds at most this many pixel centres (RND_DIRECT_PIXELS, <= 64)
constexpr uint32_t kWideListTiles = 256;      // a record touching more tiles than this AND an eighth of the frame's is not binned but put on the wide list (floors, walls, sky)
constexpr uint32_t kWideListCap = 65536;
constexpr uint32_t kRasterMaxSlices = 8;      // raster workgroups that may share one heavy tile (256 candidates per batch each)
constexpr uint32_t kRasterExtraCap = 2048;    // work items (tile, slice >= 1) per pass
constexpr uint32_t kOwnerGroup = 2;            // sort-first ownership: full-width stripes of this many tile rows, dealt round-robin (4K over 8 ranks: 34 stripes, 5 or 4 a rank; with 4 rows a stripe it was 3 or 2: the heaviest rank 0.084 ms a frame against 0.078; single rows: every rank slower)
// Stripe g belongs to rank world - 1 - g % world: the stripes are dealt from the LAST rank, so that when they do not divide
// evenly the larger shares go to the high ranks and rank 0, which also assembles the frame, keeps a smaller one.
__host__ __device__ inline uint32_t stripe_owner(uint32_t group, uint32_t world) { return world - 1u - group % world; }
constexpr int kSubpixelBits = 8;
constexpr int kSub = 1 << kSubpixelBits;       // 256 sub-pixel steps per pixel
constexpr float kGuardBand = 16.0f;            // guard band half-extent in NDC units
constexpr uint32_t kMaxRenderSize = 8192;      // tile coordinates are packed in 8 bits
constexpr uint32_t kXcds = 8;                   // MI355X: workgroup b of a launch runs on XCD b % 8, each XCD with its own L2
constexpr uint64_t kClearKey = (uint64_t(0x3F800000u) << 32) | 0xFFFFFFFFull;   // depth 1.0, no triangle
constexpr uint32_t kMaxLights = 256;
constexpr uint32_t kChunkTris = 1024;          // triangles per setup work unit (one workgroup) in large frames
constexpr uint32_t kChunkTrisSmall = 256;      // ... in frames of at most kSmallFrameTris triangles: four times the workgroups,
                                               // one pass each (C1' 4150 -> 5660 frames/s, C2 9300 -> 10000; at C3's 10 M
                                               // triangles the 30 k extra units cost the cull/scan/bin kernels 4 %, C5 18 %)
constexpr uint32_t kSmallFrameTris = 2u << 20;   // measured crossover on C2-like grids: 256 wins by 2 % at 1.96 M triangles, loses by 1-2 % at 3.6 M and 5.8 M (RND_SMALL_FRAME_TRIS overrides)
constexpr uint32_t kVertexBlock = 256;         // vertices per transform workgroup
constexpr uint32_t kShadowSize = 2048;         // shadow map edge (visual_server.rs:366-367)
constexpr uint32_t kShadowCascades = 4;        // settings.shadow_cascades (visual_server.rs:88)

// flags of a draw item
enum : uint32_t {
    kItemBillboardMask = 3u,        // reference BillboardMode (material.rs:21-25)
    kItemUnlit = 4u,                // MaterialUniform.unlit (visual_server.rs:983)
    kItemSkip = 8u,                 // alpha test fails for every fragment (base.a < 0.5, untextured)
    kItemCastsShadows = 16u,        // RenderMeshInstance.casts_shadows (visual_server.rs:961)
    kItemAlphaTest = 32u,           // the texture has alpha below 1: base.a < 0.5 is decided per fragment
};

// A run of kChunkTris (or kChunkTrisSmall) consecutive triangles of a submesh, described once at registration: the
// object-space box of its vertices and the range of vertex indices it uses.  Per frame the box is
// projected to decide whether the chunk can touch the view volume and the screen stripes this
// context owns (k_cull); chunks that cannot are never set up, and vertex blocks no surviving chunk
// uses are never transformed.
struct ChunkInfo {
    float lo[3], hi[3];
    uint32_t vmin, vmax;
};
static_assert(sizeof(ChunkInfo) == 32, "ChunkInfo is 2 x 16 bytes");

// One (instance, submesh) pair of VisualServer::render's draw list (visual_server.rs:225-245).
struct alignas(16) DrawItem {
    const float* px;            // SoA object-space positions, read by the vertex-transform kernel
    const float* py;
    const float* pz;
    const rnd_vertex* verts;    // the reference's 32-byte AoS vertices, gathered when shading
    const uint32_t* indices;    // u32 TriangleList
    uint32_t n_verts;
    uint32_t n_tris;
    uint32_t vert_base;         // first slot of this item in the per-frame clip-position buffer
    uint32_t tri_base;          // id of this item's first triangle (draw order)
    float world[16];            // column-major matrix applied to positions (model, or the billboard basis)
    float nrm[16];              // column-major matrix applied to normals (model, or identity)
    float fixed_clip[4];        // billboard FixedSize: projected + divided anchor (render_mesh.wgsl:69-71)
    float base_color[4];        // MaterialUniform.base_color
    uint32_t flags;
    uint32_t texture;           // index into the texture table, RND_NONE = white
    const ChunkInfo* chunks;    // n_chunks boxes of this submesh
    uint32_t n_chunks;
    uint32_t unit_base;         // index of this item's first chunk among the frame's work units
};
static_assert(sizeof(DrawItem) == 240, "DrawItem is copied in 16-byte pieces");

// LightUniform (visual_server.rs:987-996) reduced to what fs_main_blinn_phong reads.
struct GpuLight {
    float pos[3];       // transform.w.xyz (point lights)
    float radius;
    float dir[3];       // transform.z.xyz (directional lights), not normalized, as the shader
    uint32_t kind;
    float color[4];     // rgb + intensity
    int32_t shadow_index;   // index of this light's cascade set, -1 = none
    float inv_radius;       // 1 / radius (point lights)
    uint32_t pad[2];
};

// Shadow cascades of one directional light as the shading kernel reads them: the three matrices the
// reference uploads (LightUniform::cascades_world_to_light, visual_server.rs:409-413) and the four
// depth layers, each the tile-major visibility words of a kShadowSize^2 depth-only pass (depth bits
// in the high half).
struct ShadowSet {
    float world_to_light[3][16];
    const uint64_t* layers[kShadowCascades];
};

// One registered texture as the shading kernel samples it: every mip level as the caller's RGBA8 bytes (sRGB-encoded
// colour, linear alpha: the reference's Rgba8UnormSrgb texture, backend.rs:365-383), decoded at the tap through a
// 512-entry table; levels packed largest first, level l being max(1, width >> l) x max(1, height >> l).
// (The font atlas of the 2D pass uses the same descriptor over RGBA f32 texels.)
struct TextureDesc {
    const void* texels;
    uint32_t width, height, mips;
    uint32_t opaque;                // every texel alpha == 1
};
static_assert(sizeof(TextureDesc) == 24, "TextureDesc layout");

// Per-frame constants passed by value to the kernels.
struct FrameConst {
    float pv[16];          // projection * view (render_mesh.wgsl:48)
    float view[16];
    float cam_pos[3];      // camera_transform.w.xyz
    uint32_t tone;
    float ambient[4];      // rgb + intensity (visual_server.rs:67)
    float clear_hdr[3];    // Color::GRUE after the f16 target
    uint32_t width, height;
    uint32_t tiles_x, tiles_y;
    uint32_t n_items, n_lights;
    uint32_t n_tris, n_verts;
    uint32_t rank, world;
    uint32_t n_units;              // setup work units of the frame (chunks over all draw items)
    uint32_t chunk_tris;           // triangles per work unit this frame: kChunkTris or kChunkTrisSmall
    uint32_t n_samples;            // 1, or 4 (MSAA): visibility planes of the frame
    int32_t sub_dx, sub_dy;        // whole sub-pixel steps added to every snapped vertex: the sample this
                                   // geometry pass resolves, moved onto the pixel centre (0 without MSAA)
    double ndc_step_x, ndc_step_y;   // 2/width, 2/height in f64 (pixel centre -> NDC at shade time)
    float ndc_step_xf, ndc_step_yf;  // the same in f32
    // Multi-view passes (the four shadow cascades of a directional light in ONE geometry pass): the views are
    // stacked vertically in one tall render target of width x (n_views * view_h); view v has its own
    // projection * view (pv for view 0, pv_more[v - 1] above), its viewport is rows [v * view_h, (v + 1) * view_h)
    // and doubles as its scissor.  A work unit is (draw item, chunk, view); n_units counts all of them,
    // n_units_view those of one view.  Single-view passes: n_views 1, view_h = height, n_units_view = n_units.
    uint32_t n_views, view_h, view_mask, n_units_view;
    float pv_more[3][16];
    // World position of a fragment from its interpolated clip x, y, w (k_shade): cam_pos + unproject * (x, y, w),
    // column-major 3x3 = camera_transform's linear part * diag(1/P00, 1/P11, 1); valid once set_camera has made the
    // projection a perspective_lh (unproject_ok), where view-space is (x / P00, y / P11, w).
    float unproject[9];
    uint32_t unproject_ok;
};
constexpr uint32_t kMaxViews = 4;
constexpr uint32_t kUnitViewShift = 28;        // active_units.y = chunk | view << 28

// Setup record of one (sub-)triangle after cull / clip / snap: 48 bytes.
struct SetupRecord {
    int32_t x0, y0, x1, y1, x2, y2;    // snapped framebuffer positions, 8 sub-pixel bits
    float z0, dz1, dz2;                // depth at v0 and deltas to v1, v2
    float inv_area;                    // 1 / (double area in sub-pixel units)
    uint32_t tri_id;                   // draw-order id of the source triangle
    uint32_t flags;                    // bit 0: bbox below 64 px in x and y (32-bit edge functions are exact); bit 1: fragments take the texture alpha test
};
static_assert(sizeof(SetupRecord) == 48, "SetupRecord is 3 x 16 bytes");

// Device-side counters of one frame.
// Statistics of k_setup's workgroups (triangles that survived setup in the low half, bbox pixels of the triangles
// rasterized at setup in the high half of one 64-bit word), striped over kStatStripes words that each sit in a cache
// line of their own: the L2 serializes atomics per LINE (~88 per microsecond), so 32 words next to each other cost
// k_setup 60 % at C3, 32 lines nothing measurable.  One block of kStatStripes * kStatStride words per counter set.
constexpr uint32_t kStatStripes = 32, kStatStride = 16;
// A vertex after K1's viewport transform and snap (raster rule 4), with its clip-space verdicts, so that
// K2a's per-triangle work starts from integers: each vertex is shared by about six triangles.
struct SnapVert {
    int32_t x, y;                 // snapped position, 1/256 px (before the MSAA sample shift)
    float z;                      // z / w
    uint32_t code;                // kOut* bits
};
static_assert(sizeof(SnapVert) == 16, "SnapVert layout");
constexpr uint32_t kOutPlanes = 0x3Fu;        // bit p: outside view-volume plane p (x+w, w-x, y+w, w-y, z, w-z < 0)
constexpr uint32_t kOutNeedsClip = 0x100u;    // outside near/far or the guard band: the triangle goes to the clipper
constexpr uint32_t kOutNonFinite = 0x200u;    // a non-finite clip coordinate: the triangle is dropped
constexpr uint32_t kOutBehind = 0x400u;       // w <= 0

// ---- 2D overlay (pipeline2d.rs) --------------------------------------------------------------
// One instanced quad of the 2D pass after its vertex stage: a UI box (uibox.wgsl) or a glyph (text.wgsl).
struct Prim2D {
    int32_t xa, xb, yt, yb;       // snapped corners (1/256 px): x at quad u = 0 / 1, y at the top (v = 0) / bottom
    float color[4];               // box: pow(rgb, 2.2), alpha
    uint32_t kind;                // 0 box, 1 glyph
    uint32_t glyph;               // atlas cell 0..127
    float font_size;
    uint32_t pad;
};
static_assert(sizeof(Prim2D) == 48, "Prim2D layout");

struct FrameCounters;
// The counter sets of a frame's passes, as the kernels behind the geometry phase look at them: a frame whose geometry
// overflowed an internal stream is not presented — k_shade, k_present and k_overlay all leave the image alone.
struct OverflowGate {
    const FrameCounters* all_counters;
    uint32_t n_counter_sets;
    uint32_t no_chain;            // the frame was launched without k_clip .. k_raster: it is also withheld when a pass turns out to need them
    uint64_t ran_mask;            // bit s: counter set s was written this frame (sets beyond 63: assumed)
};

struct OverlayParams {
    const Prim2D* prims;          // boxes first, then the glyphs of every text, in draw order
    const float* srgb_lut;        // 256 entries: sRGB8 -> linear
    TextureDesc font;             // RGBA f32, one level (Rgba8Unorm)
    const uint64_t* depth_src;               // debug view: tile-major (depth bits << 32 | id) words, or null
    uint32_t n_prims;
    uint32_t fullscreen;          // 0 none, 1 depth of the 3D target, 2 a shadow map
    uint32_t src_w, src_h, src_tiles_x;
    uint32_t sw, sh, tiles_x;     // surface size and its tiles per row
    uint32_t rank, world;
    uint32_t* frame;              // row-major surface RGBA8, or null on a tile partition
    uint32_t* tiles;              // this context's compact tile-major RGBA8 on a tile partition
    uint32_t* overlay_out;        // row-major surface-sized 2D target
    OverflowGate gate;
};

struct FrameCounters {
    uint32_t n_records;        // setup records written by the clipper (after the deterministic slots)
    uint32_t n_clip;           // triangles queued for the clipper
    uint32_t n_entries;        // bin entries (sum of tile counts)
    uint32_t overflow;         // bit 0 records, bit 1 clip queue, bit 2 entries
    uint32_t n_valid;          // valid setup records (statistics)
    uint32_t n_active;         // work units that survived k_cull
    uint32_t all_verts;        // a surviving unit uses a wide vertex range: transform every vertex block
    uint32_t n_wide;           // records on the wide list (not binned: every tile's raster workgroup checks them)
    uint32_t n_extra;          // (tile, slice) work items of heavy tiles beyond their first raster workgroup
    uint32_t n_fragments;      // pixel centres in the bboxes of the triangles k_setup rasterizes itself (saturating; summed by k_scan_tiles, or by k_shade in a frame without the record chain)
    uint32_t chain_needed;     // k_setup wrote a record or queued a triangle for the clipper: the pass needs k_clip .. k_raster
    uint32_t n_far;            // early-depth frames, two passes: work units of the far items that passed k_cull (candidates of k_cull_far)
    uint32_t n_near;           // ... and the number of active units of the first pass (k_hiz notes it; k_cull_far appends the far survivors behind them)
};

// What the frame's shading kernel reports to the host through mapped pinned memory, so that the host learns of an
// overflow without synchronizing: OR / maximum over the counter sets of the frame's passes.  `frame_tag` is
// written last (after a system-scope fence); a report belongs to the frame whose tag it carries.
// overflow: bits 0..2 as FrameCounters::overflow; kReportNoChainFailed: the frame ran without the record chain, needed it and
// was withheld; kReportChainNeeded: some pass wrote records or queued triangles for the clipper.
constexpr uint32_t kReportNoChainFailed = 8u, kReportChainNeeded = 16u;
struct FrameReport {
    uint32_t overflow, n_records, n_clip, n_entries;
    uint32_t n_fragments;      // the main pass's FrameCounters::n_fragments: depth complexity = n_fragments / (pixels * samples)
    uint32_t pad[2];
    uint32_t frame_tag;
};
static_assert(sizeof(FrameReport) == 32, "FrameReport layout");

struct Capacities {
    uint32_t records;       // slots in the record stream: setup_slots + room for the clipper's records
    uint32_t clip_queue, entries;
    uint32_t setup_slots;   // slots addressed by k_setup: kChunkTris per work unit
    uint32_t wide_list;     // entries of the wide-record list
};

}  // namespace rnd
