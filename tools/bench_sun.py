"""C2/C3-class grids with a shadowed directional light and a textured material (the 'full' shading variant)."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import bench
from renderer_amd import VisualServer, scenes
from renderer_amd.types import Color, Image, Light, Material, from_rotation_x

name = sys.argv[1] if len(sys.argv) > 1 else "c2_grid_1m"
textured = len(sys.argv) > 2 and sys.argv[2] == "tex"
scene = getattr(scenes, name)()
scene.lights.append((900, from_rotation_x(1.1), Light.directional().with_color(Color(1.0, 0.9, 0.8, 2.0))))
vs = VisualServer(scene.width, scene.height, device=0)
if textured:
    rng = np.random.default_rng(1)
    img = Image(256, 256, rng.integers(0, 256, size=(256, 256, 4), dtype=np.uint8))
    img.pixels[..., 3] = 255
    img.make_mips()
    h = vs.asset_server.add(img)
    mat = vs.asset_server.add(Material(base_color_image=h))
    for m in scene.meshes:
        for sub in m.submeshes:
            sub.material = mat
drv = bench.FrameDriver(vs, scene)
for _ in range(5):
    drv.frame()
vs.finish()
vs.mean_timings(reset=True)
t0 = time.perf_counter()
steps = 60
for _ in range(steps):
    drv.frame()
vs.finish()
dt = time.perf_counter() - t0
mean, _ = vs.timed_means(drv.frame)
print(f"{name} + sun{' + texture' if textured else ''}: {steps / dt:.1f} frames/s", {k: round(v, 4) for k, v in mean.items()})
