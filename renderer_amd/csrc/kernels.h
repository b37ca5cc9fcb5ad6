// Launch interface of the HIP kernels (kernels.hip).  All launches are asynchronous on `stream`.
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.h"

namespace rnd {

struct TileRef {
    uint32_t tile;      // global tile index (ty * tiles_x + tx)
    uint32_t compact;   // position of the tile in this rank's ascending list of owned tiles
};

// The main pass's shape, by value in the shading kernel's arguments (k_shade: its first loads depend on nothing else).
struct ShadeDims {
    uint32_t n_lights, n_items;
    uint32_t width, height, tiles_x;
    uint32_t tiles_x_magic;         // floor(2^32 / tiles_x) + 1 (0 when tiles_x is 1): n / tiles_x == umulhi(n, magic) for n < 2^24
    uint32_t rank, world;           // the sort-first partition (FrameConst::rank, world)
};

struct FrameBuffers {
    const FrameConst* frame_const;  // this pass's constants, in the per-frame table block
    // per-frame scene tables
    const DrawItem* items;
    const uint32_t* vert_bases;     // n_items + 1
    const uint32_t* tri_bases;      // n_items + 1
    const GpuLight* lights;
    const ShadowSet* shadow_sets;   // indexed by GpuLight::shadow_index
    uint32_t n_shadow_sets;
    const TextureDesc* textures;    // indexed by DrawItem::texture
    const float* srgb_lut;          // 512 entries: [0,256) sRGB8 -> linear, [256,512) alpha8 -> [0,1]
    uint32_t alpha_test;            // some draw item has kItemAlphaTest: launch the raster variants with the test
    uint32_t has_textures;          // some draw item is textured: launch the full shading variant
    const uint32_t* unit_bases;     // n_items + 1 ascending first work units of the items' ranges in processing order, then the n_items draw items of those ranges
    uint2* active_units;            // (item, chunk) of the work units that survived k_cull
    uint32_t* vblock_needed;        // per kVertexBlock block of the frame's vertex slots: used by a surviving unit
    // geometry stage
    float4* clip;                   // n_verts clip positions K1 keeps for k_shade, or null: k_shade makes them again (corner_clip)
    SnapVert* snap;                 // n_verts snapped positions + clip verdicts (K1 -> K2a)
    uint32_t* wide_list;            // record slots of the wide records of this pass
    uint32_t* tile_slices;          // per tile: raster workgroups that share it (k_scan_tiles -> k_raster)
    uint2* raster_extra;            // (tile, slice) of the workgroups beyond each tile's first
    uint32_t direct_pixels;         // K2a rasterizes triangles whose bbox holds at most this many pixel centres itself
    SetupRecord* records;
    uint2* tilewords;               // per record: .x = tx0 | ty0<<8 | (w-1)<<16 | (h-1)<<24, .y = position in its bin (single-tile records)
    uint32_t* clip_queue;           // triangle ids that need the clipper
    uint32_t* chunk_count;          // per k_setup workgroup (work unit): record slots it used
    uint32_t* tile_count;           // [0,T): single-tile records per tile; [T,2T): multi-tile records per tile
    uint32_t* tile_offset;          // per tile, exclusive scan of tile_count
    uint32_t* tile_cursor;          // per tile, fill cursor of the multi-tile records
    uint32_t* entries;              // bin entries: record slots, grouped by tile
    FrameCounters* counters;
    unsigned long long* stat_stripes;     // this pass's kStatStripes * kStatStride words (device_types.h)
    const FrameCounters* main_counters;   // the main pass's counter set (k_shade reports its fragment count)
    uint32_t no_chain;              // the frame is launched without k_clip .. k_raster (a run of frames needed none of them): k_shade sums the statistics and withholds the frame if it did need them
    uint32_t early_z;               // launch the plain main pass's k_setup with the depth test before the atomic (frames with depth complexity)
    uint32_t setup_pass;            // early_z: 0 = one pass over every unit; nonzero = two passes around k_hiz (launch_setup sets 1 and 2)
    uint32_t split_units;           // early_z, two passes: units numbered below this (the nearest draw items) form pass 1
    uint32_t bbox_px_hint_x16;      // early_z, two passes: bbox pixels per input triangle (x 16) of the last frame that ran the plain variant; a unit turned away whole counts its triangles at this rate in the depth-complexity measure
    float* hiz;                     // early_z, two passes: farthest depth per 8x8 pixel block, 16 per tile
    uint2* far_units;               // early_z, two passes: (item, chunk) of the far items' units that passed k_cull (k_cull -> k_cull_far)
    uint4* unit_bounds;             // ... and per such unit its rect of 8x8 blocks (x0 | y0 << 16, x1 | y1 << 16), the bits of a depth no sample of it is nearer than, and 1 (0: no bounds, no test)
    // the counter sets of all passes of the frame (shadow cascades, main pass, MSAA samples), read by the shading
    // kernel: a frame whose geometry overflowed an internal stream is not presented (the image keeps the previous
    // frame) and the host is told through `report`
    const FrameCounters* all_counters;
    uint32_t n_counter_sets;
    uint32_t report_tag;
    uint64_t ran_mask;              // bit s: counter set s was written this frame (sets beyond 63: assumed)
    FrameReport* report;            // mapped pinned host memory (may be null)
    // pixel stage
    const TileRef* launch_order;    // owned tiles in launch order (XCD-aware)
    uint32_t n_owned;
    ShadeDims dims;
    uint64_t* vis;                  // tile-major (depth bits << 32 | triangle id), 1024 per tile; with MSAA the
                                    // plane of the sample a geometry pass resolves / of sample 0 for shading
    uint64_t vis_stride;            // words between the planes of consecutive samples
    uint8_t* frame_rgba8;           // row-major RGBA8 frame (may be null)
    uint8_t* tiles_rgba8;           // tile-major RGBA8 of owned tiles (may be null)
    uint64_t* packed;               // row-major (depth bits << 32 | rgba8) (may be null)
    float* hdr;                     // row-major f32 rgb (may be null; read-back only)
};

// host_tables/dev_tables/table_bytes (a multiple of 16): the frame's table block, copied by the first
// clear of a frame from pinned host memory to the device; null/0 on the other passes' clears
void launch_clear(const FrameBuffers& fb, const FrameConst& fc, hipStream_t stream, const void* host_tables = nullptr,
                  void* dev_tables = nullptr, size_t table_bytes = 0);
void launch_cull(const FrameBuffers& fb, const FrameConst& fc, hipStream_t stream);
void launch_transform(const FrameBuffers& fb, const FrameConst& fc, hipStream_t stream);
void launch_setup(const FrameBuffers& fb, const FrameConst& fc, const Capacities& cap, hipStream_t stream);
void launch_clip(const FrameBuffers& fb, const FrameConst& fc, const Capacities& cap, hipStream_t stream);
void launch_scan(const FrameBuffers& fb, const FrameConst& fc, const Capacities& cap, hipStream_t stream);
void launch_bin(const FrameBuffers& fb, const FrameConst& fc, const Capacities& cap, hipStream_t stream);
void launch_raster(const FrameBuffers& fb, const FrameConst& fc, const Capacities& cap, hipStream_t stream);
void launch_shade(const FrameBuffers& fb, const FrameConst& fc, hipStream_t stream);
void shade_profile_dump();     // development aid (-DRND_SHADE_PROFILE builds; a no-op otherwise)

// read-back helpers: tile-major vis -> row-major planes (any pointer may be null)
void launch_extract(const uint64_t* vis, const TileRef* order, uint32_t n_owned, const FrameConst& fc,
                    float* depth, uint32_t* ids, uint8_t* coverage, hipStream_t stream);
// AoS vertices -> SoA positions at mesh registration
void launch_split_positions(const rnd_vertex* verts, uint32_t n, float* px, float* py, float* pz,
                            hipStream_t stream);
// tile-major RGBA8 of `n` tiles (list `tiles`) -> row-major frame
void launch_detile(const uint8_t* tiles_rgba8, const uint32_t* tiles, uint32_t n, const FrameConst& fc,
                   uint8_t* frame_rgba8, hipStream_t stream);
// every rank's gathered tile-major buffer -> row-major frame, one launch
void launch_detile_gathered(const uint8_t* gathered, uint64_t stride_bytes, uint32_t world, const FrameConst& fc,
                            uint8_t* frame_rgba8, hipStream_t stream);
// show_texture.wgsl for a 3D target whose size differs from the surface (render_size_factor != 1):
// hdr (w x h, rgb f32 holding f16 values) sampled at the surface's pixel centres, nearest or
// linear, then tone map + sRGB into the sw x sh RGBA8 frame
void launch_present(const float* hdr, uint32_t w, uint32_t h, uint8_t* frame_rgba8, uint32_t sw, uint32_t sh,
                    bool linear, uint32_t tone, const OverflowGate& gate, hipStream_t stream);
// packed (depth | rgba8) row-major -> RGBA8 frame + tile-major vis depth
// n_tiles = tiles of the surface (32x32); workgroups of tiles another rank owns return at once
void launch_overlay(const OverlayParams& p, uint32_t n_tiles, hipStream_t stream);
// element-wise minimum of n packed strips (strip r at strips + r * stride, `count` words each) -> out
void launch_min_packed(const uint64_t* strips, uint32_t n, uint64_t stride, uint64_t* out, uint64_t count, hipStream_t stream);
void launch_unpack(const uint64_t* packed, const FrameConst& fc, uint8_t* frame_rgba8, float* depth,
                   hipStream_t stream);

}  // namespace rnd
