/*
 * rnd.h — C ABI of the MI355X-native frame hot path (vertex transform -> triangle
 * setup + tile binning -> per-tile raster with depth -> deferred material/light shading).
 *
 * This is the drop-in boundary for the reference's `VisualServer`
 * (reference: src/renderer/visual_server.rs:54-874).  The reference has no FFI of its
 * own; the seam sits where `Pipeline3d::render` + `Backend` talk to wgpu
 * (src/renderer/pipeline3d.rs:261-368, src/renderer/backend.rs:455-498).  Every entry
 * point below cites the reference method it replaces.  A Rust `extern "C"` block that
 * binds these 1:1 is in integration/rust/rnd-sys.
 *
 * Conventions
 *  - every function returns RND_OK (0) or a negative rnd_status; rnd_last_error() gives text.
 *  - matrices are column-major f32[16] exactly as glam `Mat4::to_cols_array()`
 *    (visual_server.rs:212-214, :491).  Affine transforms are passed as Mat4 with last
 *    row (0,0,0,1).
 *  - the library copies everything it is given (as the reference does at
 *    backend.rs:257-276); the caller keeps ownership of its buffers.
 *  - one context is single-threaded (`&mut self` in the reference); contexts on different
 *    devices are independent.
 *  - plain pointers and sizes only; no torch / C++ types cross this boundary.
 */
#ifndef RND_H
#define RND_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rnd_ctx rnd_ctx;

typedef enum rnd_status {
    RND_OK = 0,
    RND_ERR_INVALID = -1,      /* bad argument (null pointer, size 0, index out of range) */
    RND_ERR_UNKNOWN_ID = -2,   /* mesh / material id not registered (reference: unwrap() panic) */
    RND_ERR_DEVICE = -3,       /* HIP error; maps to wgpu::SurfaceError::Lost on the Rust side */
    RND_ERR_OOM = -4,          /* allocation failed; maps to wgpu::SurfaceError::OutOfMemory */
    RND_ERR_UNSUPPORTED = -5,  /* setting exists in the reference but is not built yet */
    RND_ERR_NO_FRAME = -6,     /* read-back before the first rnd_render */
    RND_ERR_TRUNCATED = -7     /* from rnd_render only, and only as a notice: an EARLIER frame overflowed an internal
                                * stream and was therefore not presented (its image buffer kept the frame before it);
                                * the streams have been grown and the frame of this call HAS been enqueued normally */
} rnd_status;

#define RND_NONE 0xFFFFFFFFu

/* reference: src/renderer/vertex.rs:3-9 — #[repr(C)] 32-byte AoS vertex */
typedef struct rnd_vertex {
    float position[3];
    float normal[3];
    float uv[2];
} rnd_vertex;

/* reference: src/mesh.rs:9-13 `Submesh{vertices,indices,material}`; u32 TriangleList
 * indices (pipeline3d.rs:341).  material_id == RND_NONE selects the default material
 * (visual_server.rs:672-677). */
typedef struct rnd_submesh_desc {
    const rnd_vertex* vertices;
    uint32_t n_vertices;
    const uint32_t* indices;
    uint32_t n_indices;      /* multiple of 3 */
    uint32_t material_id;
} rnd_submesh_desc;

/* reference: enum ToneMapping, visual_server.rs:1218-1223 */
enum { RND_TONE_MAPPING_NONE = 0, RND_TONE_MAPPING_REINHARD = 1 };
/* reference: LightKind::id(), src/light.rs:52-59 */
enum { RND_LIGHT_DIRECTIONAL = 0, RND_LIGHT_POINT = 1 };
/* reference: BillboardMode, src/material.rs:21-25 / visual_server.rs:716-720 */
enum { RND_BILLBOARD_OFF = 0, RND_BILLBOARD_ON = 1, RND_BILLBOARD_FIXED_SIZE = 2 };

/* Per-frame kernel timings, milliseconds: HIP events recorded on the streams the kernels are launched on, one
 * before and one after each kernel of the main pass. */
typedef struct rnd_timings {
    float transform_ms;   /* K1 vertex transform (k_transform alone) */
    float setup_ms;       /* K2a triangle setup, incl. the direct small-triangle raster (k_setup alone) */
    float scan_ms;        /* tile offset scan */
    float bin_ms;         /* K2b bin fill */
    float raster_ms;      /* K3 per-tile raster (+ the further MSAA samples' geometry passes) */
    float shade_ms;       /* K4 deferred shade + tone map (+ present / 2D overlay when they run) */
    float frame_ms;       /* first kernel start -> last kernel end */
    float shadow_ms;      /* shadow-cascade depth passes (directional lights) */
    float clear_ms;       /* visibility clear + table upload (k_clear) */
    float cull_ms;        /* work-unit culling (k_cull) */
    float clip_ms;        /* K2c clipper (k_clip) */
} rnd_timings;

/* Per-frame counters used to state algorithmic bytes. */
typedef struct rnd_frame_stats {
    uint64_t n_vertices;      /* Nv: vertices transformed this frame (sum over draw items) */
    uint64_t n_triangles;     /* Nt: input triangles this frame */
    uint64_t n_setup;         /* Nt_vis: setup records written (after cull/clip) */
    uint64_t n_bin_entries;   /* Nbin: (record, tile) pairs */
    uint64_t n_pixels;        /* P: pixels this context rasterizes (owned tiles only) */
    uint64_t n_fragments;     /* pixel centres in the bboxes of the small triangles rasterized at setup (bboxes of up to 32 pixel centres; 64 in early-depth frames): / (P * samples) ~ twice the depth complexity */
    uint32_t n_draw_items;    /* M */
    uint32_t n_lights;        /* L */
    uint32_t n_tiles;         /* tiles owned by this context */
    uint32_t overflow;        /* 1 if an internal buffer had to grow and the frame was re-run */
    uint32_t early_depth;     /* 1 if the frame ran the setup variant that tests depth before the visibility atomic */
    uint32_t n_units_hidden;   /* early-depth frames: work units (runs of 256 or 1024 triangles) turned away whole behind what the first pass drew */
} rnd_frame_stats;

/* ---- lifecycle -------------------------------------------------------------------- */

/* reference: VisualServer::new (visual_server.rs:55-154).  No window: width/height are the
 * surface size; `device` is the HIP device ordinal.  Defaults as the reference: render size
 * factor 1.0, MSAA 1, tone mapping Reinhard (:85-90), ambient (0.3,0.5,0.9,0.04) (:67),
 * default material white/lit (material.rs:10-19), unit quad mesh (:865-873).
 * Limit the reference does not have: the render size (surface size, and surface size times the render size factor) is
 * at most 8192 x 8192 pixels (tile coordinates are packed in 8 bits); larger sizes return RND_ERR_INVALID here and from
 * rnd_set_render_size / rnd_set_render_size_factor. */
int rnd_create(rnd_ctx** out, uint32_t width, uint32_t height, int device);
void rnd_destroy(rnd_ctx* ctx);
/* text of the last error on this context (or of the last failed rnd_create when ctx==NULL) */
const char* rnd_last_error(const rnd_ctx* ctx);
/* "hip-gfx950" for the product library, "cpu-oracle" for the checker in oracle/ */
const char* rnd_backend_name(void);

/* ---- settings (reference: visual_server.rs:156-190) -------------------------------- */
int rnd_render_size(const rnd_ctx* ctx, uint32_t* width, uint32_t* height);  /* :156 */
int rnd_set_render_size(rnd_ctx* ctx, uint32_t width, uint32_t height);      /* :160 */
int rnd_render_size_factor(const rnd_ctx* ctx, float* factor);               /* :165 */
int rnd_set_render_size_factor(rnd_ctx* ctx, float factor);                  /* :169 */
/* Resolution of the 3D render target: render_size * render_size_factor, truncated
 * (recreate_render_targets :621-623).  rnd_read_depth / _coverage / _triangle_ids / _color_hdr
 * return planes of this size; rnd_read_color_rgba8 returns the presented frame at render_size
 * (show_texture.wgsl over the surface: nearest sampler at factors <= 1, linear above, :629-633).
 * No counterpart in the reference (its tests read nothing back). */
int rnd_render_resolution(const rnd_ctx* ctx, uint32_t* width, uint32_t* height);
int rnd_msaa_sample_count(const rnd_ctx* ctx, uint32_t* count);              /* :174 */
int rnd_set_msaa(rnd_ctx* ctx, uint32_t sample_count);                       /* :178 */
int rnd_tone_mapping(const rnd_ctx* ctx, uint32_t* mode);                    /* :183 */
int rnd_set_tone_mapping(rnd_ctx* ctx, uint32_t mode);                       /* :187 */

/* ---- resources (reference: register_mesh :664-695, register_material :697-750) ----- */
/* Replaces the AssetServer lookup + create_vertex_buffer/create_index_buffer upload.
 * Registering an id twice replaces the mesh.  A submesh of more than 4 GiB of vertices (2^27 of them) or of indices
 * (2^30) is refused with RND_ERR_INVALID: the kernels address a submesh's arrays by 32-bit byte offsets. */
int rnd_register_mesh(rnd_ctx* ctx, uint32_t mesh_id, uint32_t n_submeshes,
                      const rnd_submesh_desc* submeshes);
/* reference: update_render_material_data :710-750 packs MaterialUniform :978-985.
 * texture_id RND_NONE = the 1x1 white texture (:736). */
int rnd_register_material(rnd_ctx* ctx, uint32_t material_id, const float base_color[4],
                          uint32_t billboard_mode, uint32_t unlit, uint32_t texture_id);
/* reference: register_texture/update_texture :752-769; data layout = Image::data(),
 * tightly packed RGBA8 mip levels, largest first, level l = max(1,w>>l) x max(1,h>>l)
 * (crates/asset_image/src/image.rs:34-113); sampled as Rgba8UnormSrgb (backend.rs:376) with
 * Repeat + trilinear (backend.rs:414-426).  A texture with any alpha below 255 turns on the
 * per-fragment alpha test (render_mesh.wgsl:89-92) for the draw items that use it. */
int rnd_register_texture(rnd_ctx* ctx, uint32_t texture_id, uint32_t width, uint32_t height,
                         uint32_t mip_levels, const uint8_t* rgba8_mipchain);

/* ---- per-frame state --------------------------------------------------------------- */
/* reference: set_camera(&Affine3A, &Camera) :206-222; Camera{vfov,aspect_ratio,near,far}
 * src/camera.rs:4-9; projection = perspective_lh (camera.rs:27-29); view = inverse(transform). */
int rnd_set_camera(rnd_ctx* ctx, const float transform_cols[16], float vfov, float aspect_ratio,
                   float z_near, float z_far);
/* reference: set_light(UniqueNodeId, Affine3A, &Light) :350-474; color.a is the intensity;
 * radius is ignored for directional lights (Light::radius().unwrap_or_default()).
 * A directional light's four shadow cascades (compute_shadow_cascade_projviews :771-863) are
 * computed here, from the camera current at this call, as create_light/update_light do
 * (:380-381, :445-446); rnd_set_camera does not recompute them (:221).
 * Limit the reference does not have: at most 256 lights per context (the reference's list is unbounded, :350-436);
 * the 257th new id returns RND_ERR_INVALID. */
int rnd_set_light(rnd_ctx* ctx, uint64_t id, const float transform_cols[16], const float color[4],
                  uint32_t kind, float radius);
/* reference: set_mesh_instance(UniqueNodeId, Affine3A, Handle<Mesh>, &AssetServer) :481-506
 * (material_override None, casts_shadows true). */
int rnd_set_mesh_instance(rnd_ctx* ctx, uint64_t id, const float transform_cols[16],
                          uint32_t mesh_id);
/* Same, exposing RenderMeshInstance{material_override, casts_shadows} (:955-962), which the
 * reference only sets through set_sprite (:539-548). */
int rnd_set_mesh_instance_ex(rnd_ctx* ctx, uint64_t id, const float transform_cols[16],
                             uint32_t mesh_id, uint32_t material_override, uint32_t casts_shadows);
/* The same for n instances in one call (the engine re-submits every node every frame,
 * engine.rs:136-143; a binding can hand the whole list over at once): ids[n], 16 floats per transform,
 * mesh_ids[n]; material_override none, casts_shadows true.  Stops at the first error.  No counterpart in
 * the reference. */
int rnd_set_mesh_instances(rnd_ctx* ctx, uint32_t n, const uint64_t* ids, const float* transforms_cols,
                           const uint32_t* mesh_ids);
/* reference: set_sprite(id, Affine3A, Handle<Image>, Color, ..) :508-550 — unit quad, billboard
 * On, unlit, no shadows, base colour = `color`, texture = texture_id (RND_NONE = white). */
int rnd_set_sprite(rnd_ctx* ctx, uint64_t id, const float transform_cols[16], uint32_t texture_id,
                   const float color[4]);
/* reference: reset_scene :579-581 — drops meshes, materials, textures, lights, instances. */
int rnd_reset_scene(rnd_ctx* ctx);

/* ---- 2D overlay (reference: Pipeline2d, src/renderer/pipeline2d.rs:266-310) ---------- */
/* The reference draws a second, surface-sized Rgba8UnormSrgb target every frame (cleared to
 * transparent; visual_server.rs:110-118, :641-651): an optional full-screen debug view of a depth
 * texture, then the UI boxes, then the texts, each alpha-blended in that order; Backend::render
 * alpha-blends it over the presented 3D image (backend.rs:470-491).  All of it is per-pixel work
 * on the final frame and runs as one more kernel after shading, only when something is set. */
/* reference: UiBoxInstance, src/renderer/pipeline2d/uibox_instance.rs:5-11 (#[repr(C)], 32 B);
 * position = top-left corner in surface pixels (y down), colour is gamma 2.2 encoded (uibox.wgsl:27) */
typedef struct rnd_uibox {
    float position[2];
    float size[2];
    float color[4];
} rnd_uibox;
/* reference: set_uiboxes(&[UiBoxInstance]) visual_server.rs:476-479 — replaces the list */
int rnd_set_uiboxes(rnd_ctx* ctx, uint32_t n, const rnd_uibox* boxes);
/* reference: set_font_image(Handle<Image>, ..) :192-204 — a wide atlas of 128 monospaced glyph
 * cells holding a signed distance field (text.wgsl:53-63, :76-105), RGBA8 sampled as Rgba8Unorm
 * (linear, one level; backend.rs:385-412) with the Repeat + linear sampler.  Default: the 1x1
 * (255,255,0,255) texture (:82). */
int rnd_set_font_image(rnd_ctx* ctx, uint32_t width, uint32_t height, const uint8_t* rgba8);
/* reference: reset_texts :552-554, add_text(NodeId, TextDescriptor{text, position, font_size})
 * :556-577 — one glyph quad per byte (ids above 127 become 127), advance font_size*1.1667*0.5 */
int rnd_reset_texts(rnd_ctx* ctx);
int rnd_add_text(rnd_ctx* ctx, const uint8_t* text, uint32_t n_bytes, float x, float y, float font_size);
/* reference: set_depth_fullscreen_texture :313-326, set_shadow_map_fullscreen_texture(light) :328-346
 * (layer 0 of the light's cascade maps; an unknown light leaves the setting as it is, as the
 * reference's warning path does), unset_fullscreen_texture :348-350; fullscreen_texture.wgsl shows
 * (2*0.5)/(50.5 - d*49.5) as grey.  rnd_reset_scene also unsets it (it lives in RenderScene, :903). */
int rnd_set_depth_fullscreen_texture(rnd_ctx* ctx);
int rnd_set_shadow_map_fullscreen_texture(rnd_ctx* ctx, uint64_t light_id);
int rnd_unset_fullscreen_texture(rnd_ctx* ctx);
/* the 2D target of the last frame, RGBA8 (sRGB-encoded colour, linear alpha), render_size; all
 * zero when nothing 2D is set.  No counterpart in the reference. */
int rnd_read_overlay_rgba8(rnd_ctx* ctx, uint8_t* dst);

/* ---- frame ------------------------------------------------------------------------- */
/* reference: VisualServer::render :224-311 (+ Pipeline3d::render, Backend::render).
 * Enqueues the whole frame on the context's stream and returns; rnd_finish waits for it.
 * Internal streams (setup records, clip queue, bin entries) are sized from the scene and grow on demand.  A frame
 * that overflows one is never shown with triangles missing: its shading kernel writes nothing (the image buffer keeps
 * the previous frame), the library learns of it without synchronizing (a report in mapped pinned memory, looked at
 * when the next frame is enqueued), grows the stream, and the next rnd_render returns RND_ERR_TRUNCATED once as a
 * notice — that call's frame is enqueued as usual.  rnd_finish (and every read-back) re-runs an overflowed frame
 * until it is complete, so what they return is always a whole frame.
 * The same notice covers one more case: after a run of frames that wrote no setup record, queued nothing for the clipper
 * and filled no bin, the four kernels of that chain are no longer launched; a frame that turns out to need them is
 * withheld in the same way (one frame keeps the previous image), the chain comes back for the next 64 frames, and
 * rnd_finish / the read-backs re-run the withheld frame in full (RND_NO_CHAIN=0 at rnd_create: always launch the chain).
 * Frames in flight may overlap inside the library (geometry of the next frame, shadow passes on streams of their own);
 * everything the caller can observe is ordered on the context's stream. */
int rnd_render(rnd_ctx* ctx);
int rnd_finish(rnd_ctx* ctx);
/* The frame of rnd_render for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc,
 * code that prepares for a HIP graph capture; the call itself waits on events and is not to be made WHILE a stream is
 * capturing).  No counterpart in the reference.  Same scene state, same kernels, same
 * pixels as rnd_render, but
 *  - the image goes to the caller's device buffer `out_rgba8_dev` (row-major width*height*4 bytes; on a tile
 *    partition the tile-major n_owned_tiles*tile_px*tile_px*4 bytes of rnd_tiles_device_ptr), and
 *  - every intermediate stream of the frame (table block, clip positions where the frame keeps them, snapped vertices,
 *    visibility words, setup records, bin entries, queues, counters) lives in one static 128 MiB device array inside the library
 *    instead of hipMalloc'ed buffers; its per-device address is looked up once in rnd_create.
 * The call allocates nothing on the device or in pinned memory and frees nothing.  Limits: the frame's streams
 * must fit the array (about 1 M triangles at 1920x1080; RND_ERR_INVALID with the sizes otherwise, before
 * anything is launched); no directional lights (their shadow cascades are 134 MB each), render_size_factor 1,
 * no 2D overlay (RND_ERR_UNSUPPORTED); the per-frame tables must fit the 1 MiB staging block of rnd_create
 * (about 4000 draw items).  A frame that overflows an internal stream is not re-run: rnd_finish reports
 * RND_ERR_OOM and the next frame is laid out with the capacities learnt.  Everything runs on the context's
 * stream, without the overlap with the previous frame.  All contexts on one GPU share the array and it keeps
 * the last frame's contents (what must start at zero is cleared by the frame's first kernel): frames through
 * this entry point must not overlap on one GPU.  A context that asks for the array while ANOTHER context's frame through
 * this entry point is still in flight is refused with RND_ERR_INVALID (and left as it was: a refused frame changes no
 * state); when it takes the array, the previous holder's frame is gone (its read-backs return RND_ERR_NO_FRAME).
 * Frames of one context follow each other on its stream. */
int rnd_render_noalloc(rnd_ctx* ctx, void* out_rgba8_dev, uint64_t out_bytes);

/* ---- read-backs (what the reference shows on the surface / in its H/J/K debug views) -- */
/* final 8-bit sRGB surface image, row-major, origin top-left, width*height*4 bytes */
int rnd_read_color_rgba8(rnd_ctx* ctx, uint8_t* dst);
/* linear HDR colour of the Rgba16Float target widened to f32, width*height*3 floats */
int rnd_read_color_hdr(rnd_ctx* ctx, float* dst);
/* Depth32Float target, width*height floats (clear value 1.0) */
int rnd_read_depth(rnd_ctx* ctx, float* dst);
/* coverage mask: 1 where a fragment won the depth test, width*height bytes */
int rnd_read_coverage(rnd_ctx* ctx, uint8_t* dst);
/* id (draw order) of the winning triangle per pixel, RND_NONE where uncovered */
int rnd_read_triangle_ids(rnd_ctx* ctx, uint32_t* dst);
int rnd_last_timings(rnd_ctx* ctx, rnd_timings* out);
/* Mean per-kernel timings over every frame enqueued since the last reset (HIP events recorded on
 * the context's stream around each kernel; harvested without stalling the frames in flight).
 * Waits for the frames still in flight.  reset != 0 clears the accumulators afterwards. */
int rnd_mean_timings(rnd_ctx* ctx, rnd_timings* mean, uint32_t* n_frames, int reset);
/* Timing events are normally recorded on a sample of the frames only (the first four, then every 256th: a timed frame runs without the
 * overlap with its neighbours, so that its kernel durations are its own).  rnd_time_frames makes the next n frames
 * timed ones; rnd_timing_stats is rnd_mean_timings with the per-kernel maximum beside the mean (either may be NULL). */
int rnd_time_frames(rnd_ctx* ctx, uint32_t n);
int rnd_timing_stats(rnd_ctx* ctx, rnd_timings* mean, rnd_timings* max, uint32_t* n_frames, int reset);
int rnd_frame_stats_get(rnd_ctx* ctx, rnd_frame_stats* out);

/* ---- multi-GPU (no counterpart in the reference, which is single-device;
 *      north_star: sort-first by screen tiles, sort-last depth composite) ------------- */
/* Sort-first: this context rasterizes/shades only the screen tiles it owns.  Ownership is
 * by full-width stripes of 2 tile rows (64 pixel rows) dealt round-robin from the last rank: owner = world - 1 - (ty / 2) mod world
 * (when the stripes do not divide evenly the larger shares go to the high ranks; rank 0 also assembles the frame). */
int rnd_set_tile_partition(rnd_ctx* ctx, uint32_t rank, uint32_t world);
/* tile edge in pixels (square tiles) and the number of tiles this context owns */
int rnd_tile_info(const rnd_ctx* ctx, uint32_t* tile_px, uint32_t* n_owned_tiles,
                  uint32_t* n_tiles_total);
/* Device pointer (on this context's device) of the tile-major RGBA8 output of the last frame:
 * n_owned_tiles * tile_px*tile_px*4 bytes, tiles in ascending tile index.  This is what a
 * rank hands to the RCCL gather. */
int rnd_tiles_device_ptr(rnd_ctx* ctx, void** dev_ptr, uint64_t* n_bytes);
/* Make this context write its tile-major output into caller-owned device memory (for example a
 * torch tensor that is then handed to the RCCL gather) instead of its own buffer; n_bytes must be
 * at least n_owned_tiles * tile_px*tile_px*4.  NULL restores the internal buffer. */
int rnd_set_tiles_output(rnd_ctx* ctx, void* dev_ptr, uint64_t n_bytes);
/* On the gathering rank: scatter `src_rank`'s tile-major buffer (device pointer on this
 * context's device) into this context's row-major RGBA8 frame. */
int rnd_detile(rnd_ctx* ctx, uint32_t src_rank, uint32_t world, const void* dev_tiles);
/* The same for all ranks at once: rank r's buffer starts at dev_gathered + r * stride_bytes (the
 * layout of a gather into one [world, stride] tensor); stride_bytes a multiple of 4 and at least
 * the largest rank's n_owned_tiles * tile_px*tile_px*4.  One launch instead of `world`. */
int rnd_detile_gathered(rnd_ctx* ctx, uint32_t world, const void* dev_gathered, uint64_t stride_bytes);
/* Sort-last: device pointer of the packed (depth bits << 32 | rgba8) u64 frame, width*height
 * entries, row-major; min-reducing these across ranks is the depth composite. */
int rnd_packed_device_ptr(rnd_ctx* ctx, void** dev_ptr, uint64_t* n_bytes);
/* Unpack a (reduced) packed frame on this context's device into its RGBA8 frame (the depth stays in the
 * packed words' upper halves). */
int rnd_unpack(rnd_ctx* ctx, const void* dev_packed);
/* Make the packed frame go to caller-owned device memory (16-byte aligned, width*height*8 bytes) instead of the
 * library's buffer, as rnd_set_tiles_output does for tiles: read when the next frame is enqueued, so two buffers may
 * alternate while an exchange still reads the other.  Turns the packed output on.  NULL restores the internal buffer. */
int rnd_set_packed_output(rnd_ctx* ctx, void* dev_ptr, uint64_t n_bytes);
/* The local step of the sort-last composite: element-wise minimum of n packed strips of `count` words each, strip r at
 * dev_strips + r*stride_words words (the layout an all-to-all leaves on the rank that owns the strip; stride even and
 * >= count, pointers 16-byte aligned), into dev_out.  Stream-ordered on the context's stream. */
int rnd_min_packed(rnd_ctx* ctx, const void* dev_strips, uint32_t n, uint64_t stride_words, void* dev_out, uint64_t count);

/* ---- the exchanges themselves, from C over RCCL (for hosts without Python: the Rust engine) ---------------------
 * RCCL is loaded by rnd_comm_init (dlopen of librccl.so.1); the library has no link-time dependency on it.  xGMI is
 * point to point, so both exchanges are groups of ncclSend / ncclRecv in which every peer uses its own link. */
#define RND_COMM_ID_BYTES 128
/* rank 0: a fresh communicator id (ncclGetUniqueId), to be handed to the other ranks by whatever channel the host has */
int rnd_comm_make_id(uint8_t id[RND_COMM_ID_BYTES]);
/* every rank (one process per GPU): join the communicator of `world` ranks on this context's device */
int rnd_comm_init(rnd_ctx* ctx, const uint8_t id[RND_COMM_ID_BYTES], uint32_t rank, uint32_t world);
int rnd_comm_destroy(rnd_ctx* ctx);
/* Sort-first, collective (every rank calls it after rnd_render; the tile partition must be the communicator's rank and
 * world): every rank sends its tile-major buffer straight to rank 0, which scatters all of them into its row-major
 * frame (rnd_color_device_ptr / rnd_read_color_rgba8 there: until the next rnd_render, read-backs on rank 0 return the
 * assembled frame).  The call first completes this rank's frame as rnd_finish does (a frame that overflowed an internal
 * stream is re-run before its tiles leave); the exchange itself is stream-ordered on the context's stream. */
int rnd_gather_tiles(rnd_ctx* ctx);
/* Sort-last, collective (every rank calls it after rnd_render of its shard with the packed output on): direct-send
 * composite — the packed frame is cut into `world` strips, strip s of every rank goes to rank s, rank s takes the
 * minimum (rnd_min_packed), the finished strips go to rank 0, which unpacks them into its frame (read-backs on rank 0
 * return that frame until the next rnd_render).  Completes this rank's shard first, as rnd_gather_tiles does. */
int rnd_composite_packed(rnd_ctx* ctx);
/* Use the caller's HIP stream (hipStream_t as void*) for this context; NULL = own stream.  Shading, present,
 * the 2D overlay, read-backs, detile/unpack and everything the caller enqueues behind rnd_render are ordered
 * on this stream.  The geometry phase of a frame runs on an internal stream that this one waits for through
 * an event (so that it overlaps the previous frame's shading); with the environment variable
 * RND_OVERLAP=0 every kernel is launched on this stream, which is what a caller needs who captures it. */
int rnd_set_stream(rnd_ctx* ctx, void* hip_stream);
/* Device pointer of the row-major RGBA8 frame (for callers that keep the image on the GPU). */
int rnd_color_device_ptr(rnd_ctx* ctx, void** dev_ptr, uint64_t* n_bytes);

#ifdef __cplusplus
}
#endif
#endif /* RND_H */
