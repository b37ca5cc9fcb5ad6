//! Hand-written `extern "C"` declarations of include/rnd.h (UNCOMPILED here: no Rust toolchain in
//! the build container).  One item per entry point; see INTEGRATION.md for how
//! `src/renderer/visual_server.rs` of the reference calls them.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct rnd_ctx {
    _private: [u8; 0],
}

pub const RND_NONE: u32 = 0xFFFF_FFFF;
pub const RND_OK: c_int = 0;
pub const RND_ERR_INVALID: c_int = -1;
pub const RND_ERR_UNKNOWN_ID: c_int = -2;
pub const RND_ERR_DEVICE: c_int = -3; // -> wgpu::SurfaceError::Lost
pub const RND_ERR_OOM: c_int = -4; // -> wgpu::SurfaceError::OutOfMemory
pub const RND_ERR_UNSUPPORTED: c_int = -5;
pub const RND_ERR_NO_FRAME: c_int = -6;
pub const RND_ERR_TRUNCATED: c_int = -7; // notice from rnd_render: an earlier frame was dropped, this one is enqueued

/// Layout-identical to `renderer::Vertex` (src/renderer/vertex.rs:3-9).
#[repr(C)]
#[derive(Clone, Copy)]
pub struct rnd_vertex {
    pub position: [f32; 3],
    pub normal: [f32; 3],
    pub uv: [f32; 2],
}

/// reference: UiBoxInstance, src/renderer/pipeline2d/uibox_instance.rs:5-11
#[repr(C)]
#[derive(Clone, Copy)]
pub struct rnd_uibox {
    pub position: [f32; 2],
    pub size: [f32; 2],
    pub color: [f32; 4],
}

#[repr(C)]
pub struct rnd_submesh_desc {
    pub vertices: *const rnd_vertex,
    pub n_vertices: u32,
    pub indices: *const u32,
    pub n_indices: u32,
    pub material_id: u32,
}

#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct rnd_timings {
    pub transform_ms: f32,
    pub setup_ms: f32,
    pub scan_ms: f32,
    pub bin_ms: f32,
    pub raster_ms: f32,
    pub shade_ms: f32,
    pub frame_ms: f32,
    pub shadow_ms: f32,
    pub clear_ms: f32,
    pub cull_ms: f32,
    pub clip_ms: f32,
}

#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct rnd_frame_stats {
    pub n_vertices: u64,
    pub n_triangles: u64,
    pub n_setup: u64,
    pub n_bin_entries: u64,
    pub n_pixels: u64,
    pub n_fragments: u64,
    pub n_draw_items: u32,
    pub n_lights: u32,
    pub n_tiles: u32,
    pub overflow: u32,
    pub early_depth: u32,
    pub n_units_hidden: u32,
}

extern "C" {
    pub fn rnd_create(out: *mut *mut rnd_ctx, width: u32, height: u32, device: c_int) -> c_int;
    pub fn rnd_destroy(ctx: *mut rnd_ctx);
    pub fn rnd_last_error(ctx: *const rnd_ctx) -> *const c_char;
    pub fn rnd_backend_name() -> *const c_char;

    pub fn rnd_render_size(ctx: *const rnd_ctx, width: *mut u32, height: *mut u32) -> c_int;
    pub fn rnd_set_render_size(ctx: *mut rnd_ctx, width: u32, height: u32) -> c_int;
    pub fn rnd_render_size_factor(ctx: *const rnd_ctx, factor: *mut f32) -> c_int;
    pub fn rnd_set_render_size_factor(ctx: *mut rnd_ctx, factor: f32) -> c_int;
    pub fn rnd_render_resolution(ctx: *const rnd_ctx, width: *mut u32, height: *mut u32) -> c_int;
    pub fn rnd_msaa_sample_count(ctx: *const rnd_ctx, count: *mut u32) -> c_int;
    pub fn rnd_set_msaa(ctx: *mut rnd_ctx, sample_count: u32) -> c_int;
    pub fn rnd_tone_mapping(ctx: *const rnd_ctx, mode: *mut u32) -> c_int;
    pub fn rnd_set_tone_mapping(ctx: *mut rnd_ctx, mode: u32) -> c_int;

    pub fn rnd_register_mesh(ctx: *mut rnd_ctx, mesh_id: u32, n_submeshes: u32, submeshes: *const rnd_submesh_desc) -> c_int;
    pub fn rnd_register_material(ctx: *mut rnd_ctx, material_id: u32, base_color: *const f32, billboard_mode: u32, unlit: u32, texture_id: u32) -> c_int;
    pub fn rnd_register_texture(ctx: *mut rnd_ctx, texture_id: u32, width: u32, height: u32, mip_levels: u32, rgba8_mipchain: *const u8) -> c_int;

    pub fn rnd_set_camera(ctx: *mut rnd_ctx, transform_cols: *const f32, vfov: f32, aspect_ratio: f32, z_near: f32, z_far: f32) -> c_int;
    pub fn rnd_set_light(ctx: *mut rnd_ctx, id: u64, transform_cols: *const f32, color: *const f32, kind: u32, radius: f32) -> c_int;
    pub fn rnd_set_mesh_instance(ctx: *mut rnd_ctx, id: u64, transform_cols: *const f32, mesh_id: u32) -> c_int;
    pub fn rnd_set_mesh_instance_ex(ctx: *mut rnd_ctx, id: u64, transform_cols: *const f32, mesh_id: u32, material_override: u32, casts_shadows: u32) -> c_int;
    pub fn rnd_set_sprite(ctx: *mut rnd_ctx, id: u64, transform_cols: *const f32, texture_id: u32, color: *const f32) -> c_int;
    pub fn rnd_reset_scene(ctx: *mut rnd_ctx) -> c_int;

    pub fn rnd_render(ctx: *mut rnd_ctx) -> c_int;
    pub fn rnd_finish(ctx: *mut rnd_ctx) -> c_int;
    pub fn rnd_render_noalloc(ctx: *mut rnd_ctx, out_rgba8_dev: *mut c_void, out_bytes: u64) -> c_int;

    pub fn rnd_read_color_rgba8(ctx: *mut rnd_ctx, dst: *mut u8) -> c_int;
    pub fn rnd_read_color_hdr(ctx: *mut rnd_ctx, dst: *mut f32) -> c_int;
    pub fn rnd_read_depth(ctx: *mut rnd_ctx, dst: *mut f32) -> c_int;
    pub fn rnd_read_coverage(ctx: *mut rnd_ctx, dst: *mut u8) -> c_int;
    pub fn rnd_read_triangle_ids(ctx: *mut rnd_ctx, dst: *mut u32) -> c_int;
    pub fn rnd_last_timings(ctx: *mut rnd_ctx, out: *mut rnd_timings) -> c_int;
    pub fn rnd_mean_timings(ctx: *mut rnd_ctx, mean: *mut rnd_timings, n_frames: *mut u32, reset: c_int) -> c_int;
    pub fn rnd_time_frames(ctx: *mut rnd_ctx, n: u32) -> c_int;
    pub fn rnd_timing_stats(ctx: *mut rnd_ctx, mean: *mut rnd_timings, max: *mut rnd_timings, n_frames: *mut u32, reset: c_int) -> c_int;
    pub fn rnd_frame_stats_get(ctx: *mut rnd_ctx, out: *mut rnd_frame_stats) -> c_int;

    pub fn rnd_set_tile_partition(ctx: *mut rnd_ctx, rank: u32, world: u32) -> c_int;
    pub fn rnd_set_mesh_instances(ctx: *mut rnd_ctx, n: u32, ids: *const u64, transforms_cols: *const f32, mesh_ids: *const u32) -> c_int;
    pub fn rnd_set_uiboxes(ctx: *mut rnd_ctx, n: u32, boxes: *const rnd_uibox) -> c_int;
    pub fn rnd_set_font_image(ctx: *mut rnd_ctx, width: u32, height: u32, rgba8: *const u8) -> c_int;
    pub fn rnd_reset_texts(ctx: *mut rnd_ctx) -> c_int;
    pub fn rnd_add_text(ctx: *mut rnd_ctx, text: *const u8, n_bytes: u32, x: f32, y: f32, font_size: f32) -> c_int;
    pub fn rnd_set_depth_fullscreen_texture(ctx: *mut rnd_ctx) -> c_int;
    pub fn rnd_set_shadow_map_fullscreen_texture(ctx: *mut rnd_ctx, light_id: u64) -> c_int;
    pub fn rnd_unset_fullscreen_texture(ctx: *mut rnd_ctx) -> c_int;
    pub fn rnd_read_overlay_rgba8(ctx: *mut rnd_ctx, dst: *mut u8) -> c_int;
    pub fn rnd_tile_info(ctx: *const rnd_ctx, tile_px: *mut u32, n_owned_tiles: *mut u32, n_tiles_total: *mut u32) -> c_int;
    pub fn rnd_set_packed_output(ctx: *mut rnd_ctx, dev_ptr: *mut c_void, n_bytes: u64) -> c_int;
    pub fn rnd_min_packed(ctx: *mut rnd_ctx, dev_strips: *const c_void, n: u32, stride_words: u64, dev_out: *mut c_void, count: u64) -> c_int;
    pub fn rnd_comm_make_id(id: *mut u8) -> c_int;
    pub fn rnd_comm_init(ctx: *mut rnd_ctx, id: *const u8, rank: u32, world: u32) -> c_int;
    pub fn rnd_comm_destroy(ctx: *mut rnd_ctx) -> c_int;
    pub fn rnd_gather_tiles(ctx: *mut rnd_ctx) -> c_int;
    pub fn rnd_composite_packed(ctx: *mut rnd_ctx) -> c_int;
    pub fn rnd_tiles_device_ptr(ctx: *mut rnd_ctx, dev_ptr: *mut *mut c_void, n_bytes: *mut u64) -> c_int;
    pub fn rnd_set_tiles_output(ctx: *mut rnd_ctx, dev_ptr: *mut c_void, n_bytes: u64) -> c_int;
    pub fn rnd_detile_gathered(ctx: *mut rnd_ctx, world: u32, dev_gathered: *const c_void, stride_bytes: u64) -> c_int;
    pub fn rnd_detile(ctx: *mut rnd_ctx, src_rank: u32, world: u32, dev_tiles: *const c_void) -> c_int;
    pub fn rnd_packed_device_ptr(ctx: *mut rnd_ctx, dev_ptr: *mut *mut c_void, n_bytes: *mut u64) -> c_int;
    pub fn rnd_unpack(ctx: *mut rnd_ctx, dev_packed: *const c_void) -> c_int;
    pub fn rnd_set_stream(ctx: *mut rnd_ctx, hip_stream: *mut c_void) -> c_int;
    pub fn rnd_color_device_ptr(ctx: *mut rnd_ctx, dev_ptr: *mut *mut c_void, n_bytes: *mut u64) -> c_int;
}
