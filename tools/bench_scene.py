"""Frames/s and per-kernel times of one synthetic scene at a chosen resolution (not a bench line).
    python tools/bench_scene.py c2_grid_1m 7680 4320 [steps]
"""
import sys, time
sys.path.insert(0, "/root/repo")
import bench
from renderer_amd import VisualServer, scenes

name, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 100
scene = getattr(scenes, name)(width=w, height=h)
vs = VisualServer(w, h, device=0)
drv = bench.FrameDriver(vs, scene)
for _ in range(10):
    drv.frame()
vs.finish()
vs.mean_timings(reset=True)
t0 = time.perf_counter()
for _ in range(steps):
    drv.frame()
vs.finish()
dt = time.perf_counter() - t0
mean, n = vs.timed_means(drv.frame)
print(f"{name} {w}x{h}: {steps / dt:.1f} frames/s", {k: round(v, 4) for k, v in mean.items()}, vs.frame_stats())
