"""Frames launched without the record chain (k_clip, k_scan_tiles, k_bin_fill, k_raster: DESIGN.md 5).  After a run of
frames whose triangles all took the setup kernel's direct path the library leaves those four launches out; a frame that
then does need them must not reach the screen half-drawn: it is withheld (the image keeps the frame before), the host
gets RND_ERR_TRUNCATED once, and the chain comes back.  Frames are enqueued back to back with NO rnd_finish; after each
one the device frame is copied stream-ordered, as a presenting host would, and every such image must be a whole frame of
one of the two scenes.  Also: rnd_finish re-runs a withheld frame, the statistics of a frame without the chain are
right, and RND_NO_CHAIN=0 keeps the chain."""
import ctypes as C

import numpy as np
import pytest

from renderer_amd import VisualServer
from renderer_amd.scenes import c2_grid_1m
from tests.helpers import oracle_server
from tests.test_gpu_parity import soup_scene

pytestmark = pytest.mark.gpu
W, H = 320, 200


def scenes_and_images():
    fine = c2_grid_1m(width=W, height=H, scale=0.2)                 # 40 000 sub-pixel triangles: no record, no clipping
    coarse = soup_scene(3, W, H, n_meshes=4, n_tri=120, size=0.8)   # large triangles crossing the near plane: records, bins, clipper
    coarse.lights = [l for l in coarse.lights if l[2].kind_id() == 1]
    fine.lights = list(coarse.lights)        # (one light set for both: lights stay registered across the frames of a context)
    want = []
    for s in (fine, coarse):
        cpu = oracle_server(W, H)
        s.populate(cpu)
        cpu.render()
        want.append((cpu.read_color_rgba8(), cpu.read_triangle_ids()))
        cpu.close()
    return fine, coarse, want


def test_alternating_frames_without_finish_present_only_whole_frames():
    import torch
    hip = C.CDLL("libamdhip64.so")
    fine, coarse, want = scenes_and_images()
    stream = torch.cuda.Stream()
    # two contexts' worth of state in one: both scenes' meshes are registered once, the instances are swapped per frame
    vs = VisualServer(W, H, device=0)
    vs.set_stream(stream.cuda_stream)
    ptr, nbytes = vs.color_device_ptr()
    handles = {"fine": [vs.asset_server.add(m) for m in fine.meshes], "coarse": [vs.asset_server.add(m) for m in coarse.meshes]}
    far = np.eye(4, dtype=np.float32); far[:3, 3] = (0.0, 0.0, -1000.0)      # behind the camera: culled as a whole

    def frame(which):
        # both scenes' instances stay registered under fixed ids (draw order and ids stay put); the one not shown is moved away
        s = fine if which == "fine" else coarse
        vs.set_camera(s.camera_transform, s.camera)
        for lid, t, light in coarse.lights:
            vs.set_light(lid, t, light)
        for name, sc in (("fine", fine), ("coarse", coarse)):
            for nid, t, m in sc.instances:
                vs.set_mesh_instance((1000 if name == "coarse" else 0) + nid, t if name == which else far, handles[name][m])
        return vs.render()

    order = ["fine"] * 10 + ["coarse"] * 6 + ["fine"] * 10 + ["coarse"] * 3 + ["fine"] * 3
    shots, notices = [], 0
    with torch.cuda.stream(stream):
        for k, which in enumerate(order):
            if not frame(which):
                notices += 1
            shot = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
            assert hip.hipMemcpyAsync(C.c_void_p(shot.data_ptr()), C.c_void_p(ptr), C.c_size_t(nbytes), 3, C.c_void_p(stream.cuda_stream)) == 0
            shots.append(shot)
            if k % 2:
                stream.synchronize()          # (lets the reports arrive; a real host just keeps rendering)
    stream.synchronize()
    # the oracle's images of the two scenes with the other scene's instances parked: the same pictures as alone, but the
    # triangle ids of the coarse scene are offset by the fine scene's triangles — compare colours
    kinds = []
    for k, shot in enumerate(shots):
        img = shot.cpu().numpy().reshape(H, W, 4).astype(int)
        d = [np.abs(img - w[0].astype(int)).max() for w in want]
        assert min(d) <= 1, f"frame {k} ({order[k]}): neither scene's whole image (differences {d})"
        kinds.append("fine" if d[0] <= 1 else "coarse")
    # what was shown is the scene asked for, or (withheld frames) the one shown before
    late = [k for k in range(1, len(order)) if kinds[k] != order[k]]
    for k in late:
        assert kinds[k] == kinds[k - 1], (k, kinds, order)
    assert kinds[9] == "fine" and kinds[15] == "coarse" and kinds[25] == "fine" and kinds[-1] == "fine"
    assert notices >= 1 and late, "the run of fine frames should have left the chain out, and the first coarse frame been withheld"
    st = vs.frame_stats()       # (rnd_finish inside)
    assert st["n_triangles"] == fine.n_triangles + coarse.n_triangles
    vs.close()


def test_finish_reruns_a_withheld_frame_and_statistics_hold(monkeypatch):
    fine, coarse, want = scenes_and_images()
    for allow in ("1", "0"):
        monkeypatch.setenv("RND_NO_CHAIN", allow)
        vs = VisualServer(W, H, device=0)
        fine.populate(vs)
        for _ in range(12):
            vs.render()
            vs.finish()
        st = vs.frame_stats()
        assert (vs.read_triangle_ids() == want[0][1]).all()
        assert st["n_setup"] > 1000 and st["n_fragments"] > 1000 and st["n_bin_entries"] == 0
        ref = VisualServer(W, H, device=0)
        monkeypatch.setenv("RND_NO_CHAIN", "0")
        fine.populate(ref)
        ref.render()
        assert ref.frame_stats()["n_setup"] == st["n_setup"] and ref.frame_stats()["n_fragments"] == st["n_fragments"]
        ref.close()
        monkeypatch.setenv("RND_NO_CHAIN", allow)
        # now a frame that needs the chain, read back at once: whole
        vs.reset_scene()
        coarse.populate(vs)
        vs.render()
        assert (vs.read_triangle_ids() == want[1][1]).all()
        assert np.abs(vs.read_color_rgba8().astype(int) - want[1][0].astype(int)).max() <= 1
        vs.close()
