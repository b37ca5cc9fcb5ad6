"""Frames/s of one workload under the settings of the VisualServer surface (MSAA, size factor, light count)."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import bench
from renderer_amd import VisualServer, scenes
from renderer_amd.types import Color, Light, from_translation

name = sys.argv[1]
msaa, factor, n_lights = int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
scene = getattr(scenes, name)()
if n_lights >= 0:
    rng = np.random.default_rng(9)
    scene.lights = [(1000 + k, from_translation(tuple(rng.uniform((-10, -5, 4.75), (10, 5, 5.25)))),
                     Light.point(2.0).with_color(Color(*rng.uniform(0.2, 1, 3), 2.0))) for k in range(n_lights)]
vs = VisualServer(scene.width, scene.height, device=0)
vs.set_msaa(msaa)
vs.set_render_size_factor(factor)
drv = bench.FrameDriver(vs, scene)
for _ in range(5):
    drv.frame()
vs.finish()
vs.mean_timings(reset=True)
steps = 50
t0 = time.perf_counter()
for _ in range(steps):
    drv.frame()
vs.finish()
dt = time.perf_counter() - t0
mean, _ = vs.timed_means(drv.frame)
print(f"{name} msaa {msaa} factor {factor} lights {len(scene.lights)}: {steps / dt:.1f} frames/s", {k: round(v, 4) for k, v in mean.items()})
