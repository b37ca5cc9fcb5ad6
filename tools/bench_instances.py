"""Many small draw items: M instances of a small mesh (quad or low-poly sphere) - host and GPU cost per frame."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import bench
from renderer_amd import VisualServer
from renderer_amd.scenes import SceneDesc, uv_sphere_mesh
from renderer_amd.types import Camera, Color, Light, Mesh, from_scale, from_translation, identity

m = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
kind = sys.argv[2] if len(sys.argv) > 2 else "quad"
w, h = 1920, 1080
s = SceneDesc("instances", w, h, Camera(aspect_ratio=w / h), identity())
s.meshes.append(Mesh.quad() if kind == "quad" else uv_sphere_mesh(16, 8, 0.5))
rng = np.random.default_rng(3)
for k in range(m):
    p = rng.uniform((-8, -4.5, 4), (8, 4.5, 12))
    s.instances.append((k + 1, from_translation(tuple(p)) @ from_scale((0.2, 0.2, 0.2)), 0))
s.lights.append((50000, from_translation((0, 0, 2)), Light.point(30.0).with_color(Color(1, 1, 1, 2))))
vs = VisualServer(w, h, device=0)
drv = bench.FrameDriver(vs, s)
for _ in range(5):
    drv.frame()
vs.finish()
vs.mean_timings(reset=True)
steps = 50
t0 = time.perf_counter()
for _ in range(steps):
    drv.frame()
t1 = time.perf_counter()
vs.finish()
t2 = time.perf_counter()
mean, _ = vs.timed_means(drv.frame)
print(f"{m} {kind} instances: {steps / (t2 - t0):.1f} frames/s, host enqueue {(t1 - t0) / steps * 1e3:.3f} ms/frame", {k: round(v, 4) for k, v in mean.items()})
# the same frames with the instance list handed over in one call
import ctypes as C
ids = np.array([a[0].value for a in drv.instances], dtype=np.uint64)
cols = np.ascontiguousarray(np.stack([np.ctypeslib.as_array(a[1], shape=(16,)) for a in drv.instances]).astype(np.float32))
meshes = np.array([a[2].value for a in drv.instances], dtype=np.uint32)
t0 = time.perf_counter()
for _ in range(steps):
    vs.api.set_camera(vs._ctx, *drv.cam_args)
    for a in drv.lights:
        vs.api.set_light(vs._ctx, *a)
    vs.api.set_mesh_instances(vs._ctx, len(ids), ids.ctypes.data, cols.ctypes.data, meshes.ctypes.data)
    vs.api.render(vs._ctx)
t1 = time.perf_counter()
vs.finish()
t2 = time.perf_counter()
print(f"  batched rnd_set_mesh_instances: {steps / (t2 - t0):.1f} frames/s, host enqueue {(t1 - t0) / steps * 1e3:.3f} ms/frame")
t0 = time.perf_counter()
for _ in range(20):
    vs.api.render(vs._ctx)
t1 = time.perf_counter()
vs.finish()
print(f"  rnd_render alone (no state calls): {(t1 - t0) / 20 * 1e3:.3f} ms/frame")
