"""Python host-side mirror of the reference's `VisualServer` over the C ABI (include/rnd.h).

reference: src/renderer/visual_server.rs:54-874.  Method names, argument meaning and error
behaviour follow the reference; the wgpu plumbing behind them is replaced by the HIP library.
Failures the reference turns into panics (`unwrap()` on an unknown handle) raise RndError.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import RND_NONE, FrameStats, RndError, SubmeshDesc, Timings
from .types import (AssetServer, Camera, Color, Handle, Light, Material, Mesh, to_cols_array)


class ToneMapping:
    """visual_server.rs:1218-1223"""
    Off = 0      # `None` in the reference
    Reinhard = 1


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class VisualServer:
    def __init__(self, width, height, asset_server=None, device=0, capi=None):
        """reference: VisualServer::new(window, asset_server) visual_server.rs:55-154"""
        self.api = capi if capi is not None else _capi.load_product()
        self.asset_server = asset_server if asset_server is not None else AssetServer()
        self._ctx = C.c_void_p()
        rc = self.api.create(C.byref(self._ctx), int(width), int(height), int(device))
        if rc != 0:
            raise RndError(rc, (self.api.last_error(None) or b"").decode())
        self._mesh_ids = {}        # Handle<Mesh> -> mesh id      (RenderScene.meshes)
        self._material_ids = {}    # Handle<Material> -> id       (RenderScene.materials)
        self._texture_ids = {}     # Handle<Image> -> id          (RenderScene.textures)
        self._next_id = 1

    # -- plumbing ---------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise RndError(rc, (self.api.last_error(self._ctx) or b"").decode())

    def close(self):
        if self._ctx:
            self.api.destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def backend_name(self):
        return self.api.backend_name().decode()

    # -- settings (visual_server.rs:156-190) --------------------------------------------
    def render_size(self):
        w, h = C.c_uint32(), C.c_uint32()
        self._check(self.api.render_size(self._ctx, C.byref(w), C.byref(h)))
        return (w.value, h.value)

    def render_resolution(self):
        """Size of the 3D render target: render_size * render_size_factor (visual_server.rs:621-623)."""
        w, h = C.c_uint32(), C.c_uint32()
        self._check(self.api.render_resolution(self._ctx, C.byref(w), C.byref(h)))
        return (w.value, h.value)

    def set_render_size(self, size):
        self._check(self.api.set_render_size(self._ctx, int(size[0]), int(size[1])))

    def render_size_factor(self):
        f = C.c_float()
        self._check(self.api.render_size_factor(self._ctx, C.byref(f)))
        return f.value

    def set_render_size_factor(self, factor):
        self._check(self.api.set_render_size_factor(self._ctx, float(factor)))

    def msaa_sample_count(self):
        n = C.c_uint32()
        self._check(self.api.msaa_sample_count(self._ctx, C.byref(n)))
        return n.value

    def set_msaa(self, sample_count):
        self._check(self.api.set_msaa(self._ctx, int(sample_count)))

    def tone_mapping(self):
        m = C.c_uint32()
        self._check(self.api.tone_mapping(self._ctx, C.byref(m)))
        return m.value

    def set_tone_mapping(self, tone_mapping):
        self._check(self.api.set_tone_mapping(self._ctx, int(tone_mapping)))

    # -- registration (visual_server.rs:664-769) ---------------------------------------
    def _register_material(self, handle):
        if handle is None:
            return RND_NONE
        if handle in self._material_ids:
            return self._material_ids[handle]
        material = self.asset_server.get(handle)
        tex = RND_NONE
        if material.base_color_image is not None:
            tex = self._register_texture(material.base_color_image)
        mid = self._next_id
        self._next_id += 1
        base = np.asarray(material.base_color.to_array(), dtype=np.float32)
        self._check(self.api.register_material(self._ctx, mid, _fp(base), int(material.billboard_mode),
                                               int(bool(material.unlit)), tex))
        self._material_ids[handle] = mid
        return mid

    def _register_texture(self, handle):
        if handle in self._texture_ids:
            return self._texture_ids[handle]
        image = self.asset_server.get(handle)
        tid = self._next_id
        self._next_id += 1
        data = np.ascontiguousarray(image.data, dtype=np.uint8)
        self._check(self.api.register_texture(self._ctx, tid, int(image.width), int(image.height),
                                              int(image.mip_level_count), data.ctypes.data))
        self._texture_ids[handle] = tid
        return tid

    def register_mesh(self, handle):
        if handle in self._mesh_ids:
            return self._mesh_ids[handle]
        mesh = self.asset_server.get(handle)
        descs = (SubmeshDesc * len(mesh.submeshes))()
        keep = []
        for d, s in zip(descs, mesh.submeshes):
            keep.append((s.vertices, s.indices))
            d.vertices = s.vertices.ctypes.data
            d.n_vertices = len(s.vertices)
            d.indices = s.indices.ctypes.data
            d.n_indices = len(s.indices)
            d.material_id = self._register_material(s.material)
        mid = self._next_id
        self._next_id += 1
        self._check(self.api.register_mesh(self._ctx, mid, len(mesh.submeshes), descs))
        self._mesh_ids[handle] = mid
        return mid

    def notify_asset_changes(self, changed_images, asset_server=None):
        """visual_server.rs:583-619: every changed image that is registered is uploaded again under its
        handle (update_texture :760-769), then every material that uses one has its data rebuilt
        (update_render_material_data :710-750).  `changed_images`: iterable of Handle<Image>."""
        if asset_server is not None:
            self.asset_server = asset_server
        changed = [h for h in changed_images if h in self._texture_ids]
        for h in changed:
            tid = self._texture_ids.pop(h)
            image = self.asset_server.get(h)
            data = np.ascontiguousarray(image.data, dtype=np.uint8)
            self._check(self.api.register_texture(self._ctx, tid, int(image.width), int(image.height),
                                                  int(image.mip_level_count), data.ctypes.data))
            self._texture_ids[h] = tid
        for mh, mid in list(self._material_ids.items()):
            material = self.asset_server.get(mh)
            if material.base_color_image in changed:
                base = np.asarray(material.base_color.to_array(), dtype=np.float32)
                self._check(self.api.register_material(self._ctx, mid, _fp(base), int(material.billboard_mode),
                                                       int(bool(material.unlit)), self._texture_ids[material.base_color_image]))

    # -- per-frame state ---------------------------------------------------------------
    def set_camera(self, transform, camera: Camera):
        """visual_server.rs:206-222"""
        t = to_cols_array(transform)
        self._check(self.api.set_camera(self._ctx, _fp(t), camera.vfov, camera.aspect_ratio,
                                        camera.near, camera.far))

    def set_light(self, id, transform, light: Light):
        """visual_server.rs:350-356"""
        t = to_cols_array(transform)
        col = np.asarray(light.color.to_array(), dtype=np.float32)
        self._check(self.api.set_light(self._ctx, int(id), _fp(t), _fp(col), light.kind_id(),
                                       float(light.radius() or 0.0)))

    def set_mesh_instance(self, id, transform, mesh_handle: Handle, asset_server=None, material_override=None,
                          casts_shadows=True):
        """visual_server.rs:481-506; the two keyword arguments expose RenderMeshInstance's other fields
        (:955-962), which the reference sets itself (None / true, and false for sprites)."""
        if asset_server is not None:
            self.asset_server = asset_server
        mid = self.register_mesh(mesh_handle)
        t = to_cols_array(transform)
        if material_override is None and casts_shadows:
            self._check(self.api.set_mesh_instance(self._ctx, int(id), _fp(t), mid))
        else:
            mat = RND_NONE if material_override is None else int(material_override)
            self._check(self.api.set_mesh_instance_ex(self._ctx, int(id), _fp(t), mid, mat, 1 if casts_shadows else 0))

    def set_mesh_instances(self, ids, transforms, mesh_handles):
        """n instances in one call (rnd_set_mesh_instances): ids (n,), transforms (n, 4, 4) as the
        per-instance call takes them, one mesh handle or a sequence of n."""
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        n = len(ids)
        cols = np.ascontiguousarray(np.stack([to_cols_array(t) for t in transforms]).astype(np.float32)) if n else np.zeros((0, 16), np.float32)
        if isinstance(mesh_handles, Handle):
            meshes = np.full(n, self.register_mesh(mesh_handles), dtype=np.uint32)
        else:
            meshes = np.ascontiguousarray([self.register_mesh(h) for h in mesh_handles], dtype=np.uint32)
        self._check(self.api.set_mesh_instances(self._ctx, n, ids.ctypes.data if n else None, cols.ctypes.data if n else None,
                                                meshes.ctypes.data if n else None))

    def set_sprite(self, id, transform, image_handle, base_color: Color, asset_server=None):
        """visual_server.rs:508-550"""
        if asset_server is not None:
            self.asset_server = asset_server
        tex = RND_NONE if image_handle is None else self._register_texture(image_handle)
        t = to_cols_array(transform)
        col = np.asarray(base_color.to_array(), dtype=np.float32)
        self._check(self.api.set_sprite(self._ctx, int(id), _fp(t), tex, _fp(col)))

    def reset_scene(self):
        """visual_server.rs:579-581"""
        self._check(self.api.reset_scene(self._ctx))
        self._mesh_ids.clear()
        self._material_ids.clear()
        self._texture_ids.clear()

    # -- 2D overlay (pipeline2d.rs) ------------------------------------------------------
    def set_uiboxes(self, uiboxes):
        """visual_server.rs:476-479; uiboxes: iterable of (position xy, size xy, Color or rgba)"""
        rows = []
        for pos, size, color in uiboxes:
            col = color.to_array() if hasattr(color, "to_array") else color
            rows.append([pos[0], pos[1], size[0], size[1], col[0], col[1], col[2], col[3]])
        arr = np.ascontiguousarray(np.asarray(rows, dtype=np.float32).reshape(-1, 8))
        self._check(self.api.set_uiboxes(self._ctx, len(rows), arr.ctypes.data if len(rows) else None))

    def set_font_image(self, image_handle, asset_server=None):
        """visual_server.rs:192-204"""
        if asset_server is not None:
            self.asset_server = asset_server
        image = self.asset_server.get(image_handle)
        data = np.ascontiguousarray(image.data, dtype=np.uint8)
        self._check(self.api.set_font_image(self._ctx, int(image.width), int(image.height), data.ctypes.data))

    def reset_texts(self):
        """visual_server.rs:552-554"""
        self._check(self.api.reset_texts(self._ctx))

    def add_text(self, id, text, position, font_size):
        """visual_server.rs:556-577 (TextDescriptor{text, position, font_size}); `id` is unused there too"""
        raw = text.encode("latin-1") if isinstance(text, str) else bytes(text)
        buf = np.frombuffer(raw, dtype=np.uint8) if raw else np.zeros(0, np.uint8)
        self._check(self.api.add_text(self._ctx, buf.ctypes.data if len(buf) else None, len(buf),
                                      float(position[0]), float(position[1]), float(font_size)))

    def set_depth_fullscreen_texture(self):
        """visual_server.rs:313-326"""
        self._check(self.api.set_depth_fullscreen_texture(self._ctx))

    def set_shadow_map_fullscreen_texture(self, light_id):
        """visual_server.rs:328-346"""
        self._check(self.api.set_shadow_map_fullscreen_texture(self._ctx, int(light_id)))

    def unset_fullscreen_texture(self):
        """visual_server.rs:348-350"""
        self._check(self.api.unset_fullscreen_texture(self._ctx))

    # -- frame ---------------------------------------------------------------------------
    def render(self):
        """visual_server.rs:224-311 — enqueues the frame; read-backs wait for it.  Returns False when the library
        gives notice (RND_ERR_TRUNCATED) that an earlier frame overflowed an internal stream and was not presented;
        this call's frame is enqueued either way."""
        rc = self.api.render(self._ctx)
        if rc == -7:
            return False
        self._check(rc)
        return True

    def finish(self):
        self._check(self.api.finish(self._ctx))

    def render_noalloc(self, out_dev_ptr, out_bytes):
        """rnd_render_noalloc: the frame into the caller's device buffer, intermediates in the library's static
        scratch array (no allocation; no counterpart in the reference)."""
        self._check(self.api.render_noalloc(self._ctx, C.c_void_p(out_dev_ptr), int(out_bytes)))

    # -- read-backs ----------------------------------------------------------------------
    def _read(self, fn, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        self._check(fn(self._ctx, out.ctypes.data))
        return out

    def read_color_rgba8(self):
        w, h = self.render_size()
        return self._read(self.api.read_color_rgba8, (h, w, 4), np.uint8)

    def read_overlay_rgba8(self):
        w, h = self.render_size()
        return self._read(self.api.read_overlay_rgba8, (h, w, 4), np.uint8)

    def read_color_hdr(self):
        w, h = self.render_resolution()
        return self._read(self.api.read_color_hdr, (h, w, 3), np.float32)

    def read_depth(self):
        w, h = self.render_resolution()
        return self._read(self.api.read_depth, (h, w), np.float32)

    def read_coverage(self):
        w, h = self.render_resolution()
        return self._read(self.api.read_coverage, (h, w), np.uint8)

    def read_triangle_ids(self):
        w, h = self.render_resolution()
        return self._read(self.api.read_triangle_ids, (h, w), np.uint32)

    def last_timings(self):
        t = Timings()
        self._check(self.api.last_timings(self._ctx, C.byref(t)))
        return t.as_dict()

    def mean_timings(self, reset=False):
        t, n = Timings(), C.c_uint32()
        self._check(self.api.mean_timings(self._ctx, C.byref(t), C.byref(n), int(reset)))
        return t.as_dict(), n.value

    def time_frames(self, n):
        """rnd_time_frames: the next n frames record per-kernel timing events (and run un-overlapped)."""
        self._check(self.api.time_frames(self._ctx, int(n)))

    def timed_means(self, render_frame, n=8):
        """Per-kernel means over n further frames rendered by `render_frame()` with timing events on every one
        (the library itself samples only every 256th frame: a timed frame runs un-overlapped and costs ~1.5 frames)."""
        self.finish()
        self.timing_stats(reset=True)
        self.time_frames(n)
        for _ in range(n):
            render_frame()
        self.finish()
        mean, _, k = self.timing_stats()
        return mean, k

    def timing_stats(self, reset=False):
        """(mean, max, n_frames) per kernel over the timed frames since the last reset."""
        mean, mx, n = Timings(), Timings(), C.c_uint32()
        self._check(self.api.timing_stats(self._ctx, C.byref(mean), C.byref(mx), C.byref(n), int(reset)))
        return mean.as_dict(), mx.as_dict(), n.value

    def frame_stats(self):
        s = FrameStats()
        self._check(self.api.frame_stats_get(self._ctx, C.byref(s)))
        return s.as_dict()

    # -- multi-GPU -----------------------------------------------------------------------
    def set_tile_partition(self, rank, world):
        self._check(self.api.set_tile_partition(self._ctx, int(rank), int(world)))

    def tile_info(self):
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._check(self.api.tile_info(self._ctx, C.byref(a), C.byref(b), C.byref(c)))
        return {"tile_px": a.value, "n_owned": b.value, "n_total": c.value}

    def _devptr(self, fn):
        p, n = C.c_void_p(), C.c_uint64()
        self._check(fn(self._ctx, C.byref(p), C.byref(n)))
        return p.value, n.value

    def tiles_device_ptr(self):
        return self._devptr(self.api.tiles_device_ptr)

    def packed_device_ptr(self):
        return self._devptr(self.api.packed_device_ptr)

    def color_device_ptr(self):
        return self._devptr(self.api.color_device_ptr)

    def set_tiles_output(self, dev_ptr, n_bytes):
        self._check(self.api.set_tiles_output(self._ctx, C.c_void_p(dev_ptr), int(n_bytes)))

    def detile(self, src_rank, world, dev_ptr):
        self._check(self.api.detile(self._ctx, int(src_rank), int(world), C.c_void_p(dev_ptr)))

    def detile_gathered(self, world, dev_ptr, stride_bytes):
        self._check(self.api.detile_gathered(self._ctx, int(world), C.c_void_p(dev_ptr), int(stride_bytes)))

    def unpack(self, dev_ptr):
        self._check(self.api.unpack(self._ctx, C.c_void_p(dev_ptr)))

    def set_packed_output(self, dev_ptr, n_bytes):
        self._check(self.api.set_packed_output(self._ctx, C.c_void_p(dev_ptr), int(n_bytes)))

    def min_packed(self, strips_ptr, n, stride_words, out_ptr, count):
        """rnd_min_packed: the local step of the sort-last composite, on this context's stream."""
        self._check(self.api.min_packed(self._ctx, C.c_void_p(strips_ptr), int(n), int(stride_words), C.c_void_p(out_ptr), int(count)))

    def comm_make_id(self):
        buf = np.zeros(128, np.uint8)
        rc = self.api.comm_make_id(buf.ctypes.data)
        if rc != 0:
            raise RndError(rc, (self.api.last_error(None) or b"").decode())
        return buf

    def comm_init(self, comm_id, rank, world):
        buf = np.ascontiguousarray(comm_id, dtype=np.uint8)
        self._check(self.api.comm_init(self._ctx, buf.ctypes.data, int(rank), int(world)))

    def comm_destroy(self):
        self._check(self.api.comm_destroy(self._ctx))

    def gather_tiles(self):
        self._check(self.api.gather_tiles(self._ctx))

    def composite_packed(self):
        self._check(self.api.composite_packed(self._ctx))

    def set_stream(self, hip_stream):
        self._check(self.api.set_stream(self._ctx, C.c_void_p(hip_stream)))
