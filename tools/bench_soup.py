"""Frames/s and per-kernel times of a random triangle soup (large triangles: the binned path).
    python tools/bench_soup.py width height n_meshes n_tri size
"""
import sys, time
sys.path.insert(0, "/root/repo")
from renderer_amd import VisualServer
from tests.test_gpu_parity import soup_scene

w, h, nm, nt, size = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
s = soup_scene(7, w, h, n_meshes=nm, n_tri=nt, size=size, lights=4)
s.lights = [l for l in s.lights if l[2].kind_id() == 1]
vs = VisualServer(w, h, device=0)
handles = s.populate(vs)
for _ in range(5):
    vs.render()
vs.finish()
vs.mean_timings(reset=True)
t0 = time.perf_counter()
n = 50
for _ in range(n):
    vs.render()
vs.finish()
dt = time.perf_counter() - t0
mean, _ = vs.timed_means(vs.render)
print(f"soup {w}x{h} {nm}x{nt} size {size}: {n / dt:.1f} frames/s", {k: round(v, 4) for k, v in mean.items()}, vs.frame_stats())
