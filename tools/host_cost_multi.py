"""Host cost per frame of bench.py's sort-first loop at N > 1, measured in one process: the share of rank r of w of C3
(`--as-rank`-style partition), the loop's events and output switching, and the collective replaced by (a) nothing,
(b) a torch.distributed gather on a single-rank NCCL group (the dispatcher's and RCCL's launch cost, no peers).
Run it with RND_TUNE=0 and RND_OVERLAP=0 / 1: with the launch mode being measured at run time the levels are not comparable.
Round 3, share of rank 3 of 8 of C3, one stream: 70-77 us a frame without the exchange stream, 82 with its wait and record,
87 with the gather; an asynchronous gather on the process group's own stream (no stream of the bench's) was no better: 92.
    python tools/host_cost_multi.py [r/w]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench
from renderer_amd import VisualServer, scenes
from renderer_amd.sort_first import TileGather

r, w = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "3/8").split("/"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
scene = scenes.c3_grid_10m()
order = os.environ.get("STREAM_ORDER", "after")
if order == "before":
    stream, comm = torch.cuda.Stream(), torch.cuda.Stream()
vs = VisualServer(scene.width, scene.height, device=0)
if order != "before":
    stream, comm = torch.cuda.Stream(), torch.cuda.Stream()
vs.set_stream(stream.cuda_stream)
vs.set_tile_partition(r, w)
info = vs.tile_info()
gathers = [TileGather(0, 1, info["n_owned"], info["tile_px"], "cuda") for _ in range(2)]
rendered = [torch.cuda.Event() for _ in range(2)]
collected = [torch.cuda.Event() for _ in range(2)]
tile_args = [(g.tiles.data_ptr(), g.tiles.numel()) for g in gathers]
torch.cuda.set_stream(comm)
drv = bench.FrameDriver(vs, scene)


def loop(n, level):
    # level 0: frames only; 1: + alternating output buffers; 2: + an event recorded after each frame; 3: + the exchange
    # stream waits for it and records its own; 4: + the frame stream waits for the exchange stream's event of two frames ago
    # (the bench loop without the collective); 5: + the collective
    for i in range(n):
        b = i & 1
        if level >= 4: stream.wait_event(collected[b])
        if level >= 1: vs.set_tiles_output(*tile_args[b])
        drv.frame()
        if level >= 2: rendered[b].record(stream)
        if level >= 3:
            comm.wait_event(rendered[b])
            if level >= 5: gathers[b].gather()
            collected[b].record(comm)


for level in range(6):
    loop(100, level)
    vs.finish(); torch.cuda.synchronize()
    n = 400
    t0 = time.perf_counter()
    loop(n, level)
    t1 = time.perf_counter()
    vs.finish(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("level %d: host enqueue %.1f us per frame, wall %.1f us per frame" % (level, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6), flush=True)
dist.destroy_process_group()
