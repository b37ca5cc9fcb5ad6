"""Where the time of a short bench run goes (the driver runs bench.py --steps 20 --warmup 5): enqueue loop, the
wait for the last frame, and what rnd_finish does after it."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import bench
from renderer_amd import VisualServer, scenes

scene = scenes.c3_grid_10m()
vs = VisualServer(scene.width, scene.height, device=0)
stream = torch.cuda.Stream()
vs.set_stream(stream.cuda_stream)
drv = bench.FrameDriver(vs, scene)
for _ in range(5):
    drv.frame()
vs.finish(); torch.cuda.synchronize()
for n in (20, 20, 200):
    t0 = time.perf_counter()
    for _ in range(n):
        drv.frame()
    t1 = time.perf_counter()
    stream.synchronize()
    t2 = time.perf_counter()
    vs.finish()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f"{n} frames: enqueue {1e3 * (t1 - t0):.3f} ms, wait for the GPU {1e3 * (t2 - t1):.3f}, rnd_finish {1e3 * (t3 - t2):.3f}, "
          f"device sync {1e3 * (t4 - t3):.3f}; per frame {1e3 * (t4 - t0) / n:.4f} ms")

# completion time of every frame of a 24-frame burst after an idle device (events on the context's stream)
for rep in range(2):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(25)]
    with torch.cuda.stream(stream):
        evs[0].record(stream)
        for k in range(24):
            drv.frame()
            evs[k + 1].record(stream)
    stream.synchronize(); vs.finish()
    print("frame-to-frame ms:", " ".join(f"{evs[k].elapsed_time(evs[k + 1]):.3f}" for k in range(24)))
