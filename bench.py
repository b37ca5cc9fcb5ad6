#!/usr/bin/env python3
"""Frame-throughput bench of the HIP hot path (contract: see DESIGN.md "Measurement").

    python bench.py --gpus N --steps K --warmup W

A step is one frame: the reference's per-frame call sequence (set_camera, set_light x L,
set_mesh_instance x M, render — engine.rs:100-165, visual_server.rs:224-311) over a synthetic
scene whose meshes are already resident in HBM.  With N > 1 the frame is partitioned sort-first by
screen tiles, one process per GPU, and the finished tiles are gathered to rank 0 over RCCL and
de-tiled there (strong scaling: the frame is fixed, the tiles are split).  The instanced workload (c5) is
partitioned sort-last instead: every rank renders its share of the instances at full resolution into a packed
(depth | rgba8) frame, and the frames are depth-composited by a direct-send exchange (strips all-to-all, local
minimum, strips to rank 0).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak (MI355X_MICROARCH.md, chip-level parameters)
N_SIMDS, VALU_CLOCK_HZ = 256 * 4, 2.4e9      # a wave64 VALU instruction takes a SIMD's 32 lanes for 2 cycles (MI355X_MICROARCH.md)
TIMED_FRAMES = 16          # frames with events around every kernel, before the warm-up and the timed steps (outside `elapsed`)
SETTLE_FRAMES = 200        # untimed frames before them: the VALU-bound kernels take ~200 frames
                           # (60 ms) of load to reach the duration they keep (k_shade at C3: 0.179 ms after 25 frames,
                           # 0.167 after 65, 0.163 after 205 and after 805; the bandwidth-bound k_transform does not
                           # move), and the per-kernel table and the roofline entries describe the kernels, not the
                           # GPU's clock ramp.  `value` / `ms_per_step` are not touched by this: they are what the
                           # K timed steps after the W warm-up steps took.

WORKLOADS = {
    # name -> (scene builder name, description)
    "c1": ("c1_sphere", "C1' UV sphere 16k tri, 1 directional light, 1280x720"),
    "c2": ("c2_grid_1m", "C2 1M-triangle grid, 8 point lights, 1920x1080"),
    "c3": ("c3_grid_10m", "C3 10M-triangle scene (16 meshes), 32 point lights, 3840x2160"),
    "c4": ("c4_layers_50m", "C4 50M triangles in 10 layers, 8 point lights, 7680x4320"),
    "c5": ("c5_instanced_100m", "C5 1000 instances x 100k triangles, 8 point lights, 3840x2160"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--scale", type=float, default=1.0, help="linear mesh-density scale (debug only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--lights", type=int, default=-1, help="keep only the first N lights (debug only)")
    ap.add_argument("--partition", default="auto", choices=["auto", "sort-first", "sort-last"],
                    help="how N > 1 GPUs share the frame (auto: sort-last for c5, sort-first otherwise)")
    ap.add_argument("--as-rank", default="", help="R/W: time the share of rank R of a W-way sort-first partition on this one "
                    "GPU, no gather (debug only: per-rank kernel times of the multi-GPU path)")
    return ap.parse_args()


def fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class FrameDriver:
    """Per-frame call sequence with pre-built ctypes arguments (keeps the host off the critical path)."""

    def __init__(self, vs, scene):
        from renderer_amd.types import to_cols_array
        self.vs, self.api, self.ctx = vs, vs.api, vs._ctx
        handles = [vs.asset_server.add(m) for m in scene.meshes]
        mesh_ids = [vs.register_mesh(h) for h in handles]
        cam = scene.camera
        self.cam_t = to_cols_array(scene.camera_transform)
        self.cam_args = (fp(self.cam_t), C.c_float(cam.vfov), C.c_float(cam.aspect_ratio), C.c_float(cam.near),
                         C.c_float(cam.far))
        self.dropped = 0
        self.keep, self.lights, self.instances = [], [], []
        for lid, t, light in scene.lights:
            tt = to_cols_array(t)
            col = np.asarray(light.color.to_array(), dtype=np.float32)
            self.keep += [tt, col]
            self.lights.append((C.c_uint64(lid), fp(tt), fp(col), C.c_uint32(light.kind_id()),
                                C.c_float(light.radius() or 0.0)))
        for nid, t, m in scene.instances:
            tt = to_cols_array(t)
            self.keep.append(tt)
            self.instances.append((C.c_uint64(nid), fp(tt), C.c_uint32(mesh_ids[m])))

    def frame(self):
        api, ctx = self.api, self.ctx
        rc = api.set_camera(ctx, *self.cam_args)
        for a in self.lights:
            rc |= api.set_light(ctx, *a)
        for a in self.instances:
            rc |= api.set_mesh_instance(ctx, *a)
        if rc:
            raise RuntimeError("frame state failed: " + (api.last_error(ctx) or b"").decode())
        rc = api.render(ctx)
        if rc == -7:        # RND_ERR_TRUNCATED: notice that an earlier frame was dropped (streams grew); this frame is enqueued
            self.dropped += 1
        elif rc:
            raise RuntimeError("frame failed: " + (api.last_error(ctx) or b"").decode())


def algorithmic_bytes(stats, n_visible_tris, scene=None):
    """Per-frame algorithmic HBM bytes of each kernel (DESIGN.md "Kernels").

    Source data (positions, indices, vertices) is counted once per DISTINCT mesh the frame draws: the instances of one
    mesh (C5: 1000 of one 100 k-triangle sphere) read the same 1.8 MB, which no cache-less machine model needs to fetch
    1000 times; what is per instance is what the frame writes and reads back (snapped vertices, clip positions)."""
    nv, nt = stats["n_vertices"], stats["n_triangles"]
    nrec, nbin, p = stats["n_setup"], stats["n_bin_entries"], stats["n_pixels"]
    src_nv, src_nt = nv, nt
    if scene is not None:
        used = sorted({int(m) for _, _, m in scene.instances})
        src_nv = min(nv, sum(len(sm.vertices) for m in used for sm in scene.meshes[m].submeshes))
        src_nt = min(nt, sum(len(sm.indices) // 3 for m in used for sm in scene.meshes[m].submeshes))
    keeps_clip = 2 * nv <= 3 * p                               # the host's per-frame rule (visual_server.cpp, store_clip)
    # early-depth frames turn whole work units away before their vertices are transformed or their triangles set up: what
    # those kernels have to read shrinks with the units kept (units are runs of 1024 triangles in frames above 2 Mi
    # triangles, of 256 below: FrameConst::chunk_tris)
    n_units = max(1, -(-nt // (1024 if nt > (2 << 20) else 256)))
    kept = max(0.0, 1.0 - stats.get("n_units_hidden", 0) / n_units)
    corners = min(3 * n_visible_tris, nv)
    return {
        "clear": 8 * p,                                        # visibility words of the owned tiles
        # SoA positions in, the snapped vertex (x, y, z/w, outcodes) and — in frames that keep them — clip xyzw out
        "transform": int(kept * (12 * src_nv + 16 * nv + (16 * nv if keeps_clip else 0))),
        # indices, each snapped vertex once, one 8-byte visibility word per pixel (direct small-triangle
        # path), and for the triangles that go through the bins a 48-byte record + (tile word, bin pos)
        "setup": int(kept * (12 * src_nt + 16 * nv)) + 8 * p + 56 * min(nbin, nrec) + (8 * p if stats.get("n_units_hidden", 0) else 0),      # (+ k_hiz reading the visibility words once)
        "bin": 12 * min(nbin, nrec) + 4 * nbin,               # (tile word, bin pos) + tile offset in, bin entries out
        "raster": 52 * nbin + 16 * min(p, 1024 * nbin),       # entries + records per bin entry, visibility tile in and out
        # vis in, rgba8 out, gathers: SURVEY.md 8d's K4 figure — the index triple per visible triangle and, per distinct
        # corner, the 32-byte source vertex + the 16-byte clip position (the latter only in frames that keep it; the
        # kernel makes it again from the vertex otherwise).  (Since round 2 the kernel needs only 28 of those 48 bytes —
        # clip position + normal — but the vertex's 12 normal bytes still arrive in its whole cache line)
        "shade": 8 * p + 4 * p + 12 * min(n_visible_tris, src_nt) + 32 * min(corners, src_nv) + (16 * corners if keeps_clip else 0),
    }


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    backend = os.environ.get("RND_BENCH_BACKEND", "nccl")      # "gloo": rehearsal of the N>1 path on fewer GPUs than ranks
    if backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from renderer_amd import VisualServer, scenes
    builder, desc = WORKLOADS[args.workload]
    kwargs = {"scale": args.scale} if args.workload in ("c2", "c3", "c4") and args.scale != 1.0 else {}
    scene = getattr(scenes, builder)(**kwargs)
    if args.lights >= 0:
        scene.lights = scene.lights[:args.lights]
        desc += f" (first {args.lights} lights only)"

    n_tris = scene.n_triangles
    scene_all = scene
    pending_last = [None]
    sort_last = world > 1 and (args.partition == "sort-last" or (args.partition == "auto" and args.workload == "c5"))
    if sort_last:
        import copy
        scene = copy.copy(scene_all)
        scene.instances = scene_all.instances[rank::world]      # this rank's shard of the instances (meshes are shared)
    vs = VisualServer(scene.width, scene.height, device=local_rank)
    stream = torch.cuda.Stream(device=local_rank)
    vs.set_stream(stream.cuda_stream)
    gathers, comm, strips = None, None, None
    if sort_last:
        from renderer_amd.sort_first import StripComposite
        strips = StripComposite(rank, world, scene.width * scene.height, "cuda")
        comm = torch.cuda.Stream(device=local_rank)
        ev = {k: [torch.cuda.Event() for _ in range(2)] for k in ("rendered", "exchanged", "reduced", "collected", "unpacked")}
    elif world > 1:
        # Sort-first: every rank shades the stripes it owns into a tile-major buffer; the buffers
        # are gathered on rank 0 (RCCL send/recv group, one xGMI link per peer) on a second stream
        # while the next frame renders into the other buffer of the pair.
        from renderer_amd.sort_first import TileGather
        vs.set_tile_partition(rank, world)
        info = vs.tile_info()
        gathers = [TileGather(rank, world, info["n_owned"], info["tile_px"], "cuda") for _ in range(2)]
        comm = torch.cuda.Stream(device=local_rank)
        rendered = [torch.cuda.Event() for _ in range(2)]
        collected = [torch.cuda.Event() for _ in range(2)]
        detiled = [torch.cuda.Event() for _ in range(2)]     # rank 0: the gather list of this buffer has been read
        tile_args = [(g.tiles.data_ptr(), g.tiles.numel()) for g in gathers]
        detile_args = [(C.c_void_p(g.gathered_all.data_ptr()), g.gathered_all.stride(0)) for g in gathers] if rank == 0 else None
        torch.cuda.set_stream(comm)                          # the exchange stream is torch's current stream from here on
    if args.as_rank and world == 1:
        r, w = (int(x) for x in args.as_rank.split("/"))
        vs.set_tile_partition(r, w)
        desc += f" (share of rank {r} of {w}, no gather)"
    driver = FrameDriver(vs, scene)
    frame_no = [0]

    def step_sort_last():
        # frame i renders into buffer pair b while the exchange of frame i-1 (the other pair) is still under way
        b = frame_no[0] & 1
        frame_no[0] += 1
        with torch.cuda.stream(stream):
            stream.wait_event(ev["exchanged"][b])      # the all-to-all that last read this packed frame is done
            vs.set_packed_output(strips.frames[b].data_ptr(), strips.frames[b].numel() * 8)
            driver.frame()
            ev["rendered"][b].record(stream)
        finish_composites()                            # frame i-1: its minimum and unpack queue behind frame i's kernels
        with torch.cuda.stream(comm):
            comm.wait_event(ev["rendered"][b])
            comm.wait_event(ev["reduced"][b])          # the minimum that last read recv[b] is done
            strips.exchange(b)
            ev["exchanged"][b].record(comm)
        pending.append(b)

    def finish_composites():
        while pending:
            b = pending.pop(0)
            with torch.cuda.stream(stream):
                stream.wait_event(ev["exchanged"][b])
                stream.wait_event(ev["collected"][b])  # the gather that last read mine[b] is done
                strips.reduce(vs, b)
                ev["reduced"][b].record(stream)
            with torch.cuda.stream(comm):
                comm.wait_event(ev["reduced"][b])
                comm.wait_event(ev["unpacked"][b])
                done = strips.collect(b)
                pending_last[0] = done
                ev["collected"][b].record(comm)
            if rank == 0:
                with torch.cuda.stream(stream):
                    stream.wait_event(ev["collected"][b])
                    vs.unpack(done.data_ptr())
                    ev["unpacked"][b].record(stream)

    def step():
        if world == 1:
            driver.frame()          # (the library launches on the stream it was given: no torch stream context needed)
            return
        if sort_last:
            return step_sort_last()
        b = frame_no[0] & 1
        frame_no[0] += 1
        g = gathers[b]
        # (no torch stream contexts in the loop: the library launches on the stream it was given, events take their stream
        # as an argument, and torch's current stream IS the exchange stream for the whole run — a context manager costs
        # the host several microseconds, and at eight ranks a frame is ~70 us of GPU time)
        stream.wait_event(collected[b])               # the gather that last read this buffer is done
        vs.set_tiles_output(*tile_args[b])
        driver.frame()
        rendered[b].record(stream)
        if rank == 0:
            detile_pending()                          # frame i-1, queued behind frame i's kernels
        comm.wait_event(rendered[b])
        comm.wait_event(detiled[b])
        gathered = g.gather()                         # (on torch's current stream: comm)
        collected[b].record(comm)
        if rank == 0:
            pending.append((b, gathered))

    pending = []

    def detile_pending():
        while pending:
            b, gathered = pending.pop(0)
            stream.wait_event(collected[b])
            rc = vs.api.detile_gathered(vs._ctx, world, *detile_args[b])
            if rc:
                raise RuntimeError("detile failed: " + (vs.api.last_error(vs._ctx) or b"").decode())
            detiled[b].record(stream)

    def fence():
        if sort_last:
            finish_composites()
        else:
            detile_pending()
        vs.finish()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # Per-kernel durations first, outside `elapsed`: SETTLE_FRAMES untimed frames, then TIMED_FRAMES frames of the same
    # workload with HIP events around every kernel (on the streams the kernels are launched on; a timed frame is not
    # overlapped with its neighbours).  They come BEFORE the warm-up and the timed steps since round 3: the VALU-bound
    # kernels follow the GPU's clock ramp (k_shade 0.179 ms after 25 frames of load, 0.163 after 205), and a run of
    # `--warmup 5 --steps 20` straight from an idle GPU measured that ramp (3110 frames/s at C3 against 3550 in steady
    # state) rather than the path.  The line says how many frames ran before the warm-up (`frames_before_warmup`).
    settle = SETTLE_FRAMES
    for _ in range(settle):
        step()
    fence()
    vs.timing_stats(reset=True)
    vs.time_frames(TIMED_FRAMES)
    for _ in range(TIMED_FRAMES):
        step()
    fence()
    mean, worst, n_frames = vs.timing_stats()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    stats = vs.frame_stats()
    frame_check = None
    if world > 1:
        frame_check = check_assembled_frame(vs, scene_all, sort_last, strips, pending_last, rank, local_rank)
        ok = torch.tensor([0 if frame_check is None or frame_check.startswith("ok") else 1], device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(ok)
        if int(ok.item()):
            if rank == 0:
                print("bench: the assembled frame differs from the single-GPU frame: " + str(frame_check), file=sys.stderr)
            raise SystemExit(3)
    ms_per_step = elapsed / args.steps * 1e3
    fps = args.steps / elapsed
    result = None
    if rank == 0:
        ids = vs.read_triangle_ids() if world == 1 and not args.as_rank else None
        n_visible = int(np.unique(ids).size - 1) if ids is not None else stats["n_setup"]
        if ids is not None and not (ids != 0xFFFFFFFF).any():
            raise SystemExit("bench frame is empty: refusing to report a number")
        alg = algorithmic_bytes(stats, n_visible, scene)
        kernels = {}
        for k in ("clear", "transform", "setup", "bin", "raster", "shade"):
            ms = mean[k + "_ms"]
            kernels[k] = {"ms": round(ms, 5), "ms_max": round(worst[k + "_ms"], 5), "alg_bytes": int(alg[k]),
                          "gbps": round(alg[k] / (ms * 1e-3) / 1e9, 1) if ms > 0 else None}
        for k in ("cull", "clip", "scan"):
            kernels[k] = {"ms": round(mean[k + "_ms"], 5), "ms_max": round(worst[k + "_ms"], 5)}
        dominant = max(("transform", "setup", "bin", "raster", "shade"), key=lambda k: mean[k + "_ms"])
        # HBM bytes and VALU instructions per launch from rocprofv3 PMC passes of THIS workload (tools/profile_set.sh ->
        # profiles/traffic_<workload>.json).  The file names the sha256 of the kernel source it was measured on; when that is
        # not the tree's, the counters describe other kernels and are left out of the line.
        tj, traffic_note = {}, None
        tpath = os.path.join(ROOT, "profiles", f"traffic_{args.workload}.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                sha = hashlib.sha256(open(os.path.join(ROOT, "renderer_amd", "csrc", "kernels.hip"), "rb").read()).hexdigest()
                if tj.get("workload") != args.workload or tj.get("n_gpus", 1) != world:
                    tj, traffic_note = {}, "no PMC passes of this workload and GPU count"
                elif tj.get("kernels_sha256") != sha:
                    tj, traffic_note = {}, f"profiles/traffic_{args.workload}.json was measured on another kernels.hip: traffic and valu counters dropped"
            except Exception:
                tj, traffic_note = {}, "unreadable traffic file"
        else:
            traffic_note = "no PMC passes of this workload under profiles/"

        def roofline_of(k):
            ach = kernels[k]["gbps"]
            r = {"bound": "hbm", "kernel": k, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(ach / HBM_PEAK_GBS, 4) if ach else None, "traffic": tj.get("kernels", {}).get(k)}
            insts = tj.get("valu_insts", {}).get(k)
            if insts and mean[k + "_ms"] > 0:
                # the VALU issue bound beside the HBM one: a wave64 VALU instruction takes a SIMD at least 2 cycles
                # (MI355X_MICROARCH.md; measured 2.25 for plain f32, 4.2 for the other classes: profiles/r02_valu_ops_microbench.txt),
                # so instructions x 2 cycles / (SIMDs x clock) is the least time the kernel's VALU work takes
                floor_ms = insts * 2 / (N_SIMDS * VALU_CLOCK_HZ) * 1e3
                r["roofline_valu"] = {"bound": "valu_issue", "insts_per_launch": insts, "floor_ms": round(floor_ms, 5),
                                      "frac": round(floor_ms / mean[k + "_ms"], 3)}
            return r
        result = {
            "metric": "Mtriangles/s", "value": round(n_tris * fps / 1e6, 2), "unit": "Mtriangles/s",
            "frames_per_s": round(fps, 2),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": desc if args.scale == 1.0 else f"{desc} (mesh scale {args.scale})",
                       "triangles": n_tris, "width": scene.width, "height": scene.height,
                       "lights": len(scene.lights), "draw_items": stats["n_draw_items"],
                       "partition": "single GPU" if world == 1 else
                       (f"sort-last instance shards over {world} GPUs, direct-send depth composite over RCCL" if sort_last else
                        f"sort-first screen tiles over {world} GPUs, RCCL gather to rank 0")},
            "roofline": roofline_of(dominant),
            # north_star asks for raster AND shade: at this triangle density the raster is the direct
            # small-triangle path inside k_setup (k_raster has no bin entries), so that kernel is reported too
            "roofline_setup": roofline_of("setup"),
            "roofline_raster": roofline_of("raster"),
            "traffic_note": traffic_note,
            "kernels": kernels,
            "dropped_frames": driver.dropped,
            "frame_check": frame_check,
            "gpu_frame_ms": round(mean["frame_ms"], 4), "timed_frames": n_frames, "frames_before_warmup": settle + TIMED_FRAMES,
            "frame_stats": stats,
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(scene, desc, args.cpu_threads)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


def check_assembled_frame(vs, scene_all, sort_last, strips, pending_last, rank, local_rank):
    """After the timed loop, outside `elapsed`: rank 0 renders the same frame alone (no partition, every instance) and
    compares it with the frame the ranks assembled — equal bits for the sort-first gather; for the sort-last composite
    equal depth bits everywhere and equal colours except where two shards tie in depth (there the composite takes the
    smaller colour word, a single GPU the first drawn).  Returns "ok ..." or a description of the difference (rank 0),
    None on the other ranks."""
    if rank != 0:
        return None
    import torch
    from renderer_amd import VisualServer
    from renderer_amd.sort_first import unpack_numpy
    alone = VisualServer(scene_all.width, scene_all.height, device=local_rank)
    scene_all.populate(alone)
    if sort_last:
        ptr, nbytes = alone.packed_device_ptr()          # (renders the frame with the packed output on)
        n = scene_all.width * scene_all.height
        single = torch.empty(n, dtype=torch.int64, device="cuda")
        hip = C.CDLL("libamdhip64.so")
        assert hip.hipMemcpy(C.c_void_p(single.data_ptr()), C.c_void_p(ptr), C.c_size_t(n * 8), 3) == 0
        got = pending_last[0].view(-1)[:n]
        torch.cuda.synchronize()
        d_single, c_single = unpack_numpy(single.cpu().numpy())
        d_got, c_got = unpack_numpy(got.cpu().numpy())
        alone.close()
        if not (d_single.view(np.uint32) == d_got.view(np.uint32)).all():
            return f"depth differs on {(d_single.view(np.uint32) != d_got.view(np.uint32)).sum()} pixels"
        ties = int((c_single != c_got).any(axis=-1).sum())
        if ties > 1e-4 * n:
            return f"colour differs on {ties} pixels of equal depth"
        return f"ok (depth bits equal, {ties} pixels where shards tie in depth)"
    alone.render()
    want = alone.read_color_rgba8()
    alone.close()
    got = vs.read_color_rgba8()
    if not (got == want).all():
        return f"colour differs on {(got != want).any(axis=-1).sum()} pixels"
    return "ok (equal bits)"


def cpu_baseline(scene, desc, threads):
    """The CPU oracle (a restatement, kind "port") on the same workload, bounded to a few frames."""
    from tests.helpers import oracle_server
    threads = threads or min(16, os.cpu_count() or 1)
    cpu = oracle_server(scene.width, scene.height, threads=threads)
    scene.populate(cpu)
    cpu.render()                                    # warm-up frame (page faults, allocations)
    n, t0 = 0, time.perf_counter()
    while True:
        cpu.render()
        n += 1
        dt = time.perf_counter() - t0
        if dt > 10.0 or n >= 5:
            break
    return {"value": round(scene.n_triangles * n / dt / 1e6, 3), "unit": "Mtriangles/s", "cores": threads,
            "kind": "port", "frames_per_s": round(n / dt, 4),
            "sample": f"{n} frame(s) of the same workload ({desc}) after one warm-up frame, {threads} threads"}


if __name__ == "__main__":
    main()
