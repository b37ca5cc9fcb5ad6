"""Upper bound for overlapping consecutive frames: two independent contexts (own buffers, own streams)
rendering C3 alternately on one GPU, against one context rendering the same number of frames."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import bench
from renderer_amd import VisualServer, scenes

scene = getattr(scenes, sys.argv[1])() if len(sys.argv) > 1 else scenes.c3_grid_10m()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200


def run(k):
    servers = [VisualServer(scene.width, scene.height, device=0) for _ in range(k)]
    drivers = [bench.FrameDriver(vs, scene) for vs in servers]
    for i in range(20):
        drivers[i % k].frame()
    for vs in servers:
        vs.finish()
    t0 = time.perf_counter()
    for i in range(n):
        drivers[i % k].frame()
    for vs in servers:
        vs.finish()
    dt = time.perf_counter() - t0
    print(f"{k} context(s): {n / dt:.1f} frames/s, {dt / n * 1e3:.4f} ms per frame", flush=True)
    for vs in servers:
        vs.close()


run(1)
run(2)
run(3)
