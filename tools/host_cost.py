import sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import torch
import bench
from renderer_amd import VisualServer, scenes
scene = scenes.c3_grid_10m()
vs = VisualServer(scene.width, scene.height, device=0)
if len(sys.argv) > 1:
    r, w = (int(x) for x in sys.argv[1].split("/"))
    vs.set_tile_partition(r, w)
drv = bench.FrameDriver(vs, scene)
for _ in range(20): drv.frame()
vs.finish()
n = 300
t0 = time.perf_counter()
for _ in range(n): drv.frame()
t1 = time.perf_counter()
vs.finish()
t2 = time.perf_counter()
print("host enqueue per frame %.1f us, total per frame %.1f us" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
# split: state calls vs render
api, ctx = vs.api, vs._ctx
t0 = time.perf_counter()
for _ in range(n):
    api.set_camera(ctx, *drv.cam_args)
    for a in drv.lights: api.set_light(ctx, *a)
    for a in drv.instances: api.set_mesh_instance(ctx, *a)
t1 = time.perf_counter()
print("state calls per frame %.1f us" % ((t1 - t0) / n * 1e6))
vs.finish()
