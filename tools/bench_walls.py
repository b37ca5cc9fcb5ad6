"""A few screen-filling quads (walls / floor class content) at 4K: the multi-tile record path."""
import sys, time
sys.path.insert(0, "/root/repo")
from renderer_amd import VisualServer
from renderer_amd.scenes import SceneDesc
from renderer_amd.types import Camera, Color, Light, Mesh, from_scale, from_translation, identity

w, h = 3840, 2160
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
s = SceneDesc("walls", w, h, Camera(aspect_ratio=w / h), identity())
s.meshes.append(Mesh.quad())
for k in range(n):
    s.instances.append((k + 1, from_translation((0.1 * k, 0.05 * k, 3.0 + k)) @ from_scale((30, 30, 1)), 0))
s.lights.append((50, from_translation((0, 0, 1)), Light.point(20.0).with_color(Color(1, 1, 1, 2))))
vs = VisualServer(w, h, device=0)
s.populate(vs)
for _ in range(5):
    vs.render()
vs.finish()
vs.mean_timings(reset=True)
t0 = time.perf_counter()
steps = 100
for _ in range(steps):
    vs.render()
vs.finish()
dt = time.perf_counter() - t0
mean, _ = vs.timed_means(vs.render)
print(f"{n} screen-filling quads: {steps / dt:.1f} frames/s", {k: round(v, 4) for k, v in mean.items()}, vs.frame_stats())
