"""Overflow of the internal streams (setup records of the clipper, clip queue, bin entries) must not reach the
screen.  With RND_TEST_SMALL_STREAMS=1 a context starts with streams so small that an ordinary
clipped, binned scene overflows all three.  Frames are enqueued back to back with NO rnd_finish and NO read-back
through the library; after each one the device frame is copied stream-ordered, as a presenting host would.  Every
such image must be either the untouched initial buffer (the overflowed frame was not presented) or the complete
frame (== a fresh context's, == the oracle's); the library must give notice (RND_ERR_TRUNCATED) and recover by
itself within a few frames.  Runs in a child process because the override is read at context creation."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CODE = r'''
import ctypes as C
import numpy as np
import torch
from renderer_amd import VisualServer
from tests.helpers import oracle_server
from tests.test_gpu_parity import soup_scene

hip = C.CDLL("libamdhip64.so")
w, h = 320, 200
s = soup_scene(3, w, h, n_meshes=4, n_tri=120, size=0.8)       # large triangles crossing the near plane: clipper + bins
s.lights = [l for l in s.lights if l[2].kind_id() == 1]
cpu = oracle_server(w, h); s.populate(cpu); cpu.render()
want = cpu.read_color_rgba8()

stream = torch.cuda.Stream()
vs = VisualServer(w, h, device=0)
vs.set_stream(stream.cuda_stream)
ptr, nbytes = vs.color_device_ptr()
marker = np.full((h, w, 4), 0x5A, np.uint8)                      # what the image buffer holds before the first frame
assert hip.hipMemcpy(C.c_void_p(ptr), C.c_void_p(marker.ctypes.data), C.c_size_t(nbytes), 1) == 0
s.populate(vs)
shots, notices = [], 0
with torch.cuda.stream(stream):
    for k in range(12):
        if not vs.render():
            notices += 1
        shot = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        assert hip.hipMemcpyAsync(C.c_void_p(shot.data_ptr()), C.c_void_p(ptr), C.c_size_t(nbytes), 3, C.c_void_p(stream.cuda_stream)) == 0
        shots.append(shot)
        if k >= 2:
            stream.synchronize()          # (lets the reports of the first frames arrive; a real host just keeps rendering)
stream.synchronize()
kinds = []
for k, shot in enumerate(shots):
    img = shot.cpu().numpy().reshape(h, w, 4)
    if (img == marker).all():
        kinds.append("kept")
    else:
        d = np.abs(img.astype(int) - want.astype(int))
        assert d.max() <= 1, f"frame {k}: a truncated frame was presented ({(d.max(axis=-1) > 1).sum()} pixels differ)"
        kinds.append("whole")
print(kinds, notices)
assert kinds[0] == "kept", "the first frame should overflow the tiny streams"
assert kinds[-1] == "whole" and notices >= 1
assert kinds == sorted(kinds, key=lambda v: v == "whole"), "once complete, frames stay complete"
# rnd_finish / read-backs always deliver a whole frame, also on a fresh context that has just overflowed
vs2 = VisualServer(w, h, device=0)
s.populate(vs2)
vs2.render()
assert np.abs(vs2.read_color_rgba8().astype(int) - want.astype(int)).max() <= 1
assert vs2.frame_stats()["overflow"] == 1
assert (vs2.read_triangle_ids() == cpu.read_triangle_ids()).all()
print("overflow ok")
'''


@pytest.mark.parametrize("overlay", [False, True], ids=["plain", "translucent-overlay"])
def test_truncated_frames_are_not_presented(overlay):
    """With a translucent UI box over the frame: the 2D pass blends in place, so running it over an image
    that a truncated frame left alone would composite the box a second time — the overlay and present kernels take the
    same overflow gate as the shading kernel."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = CODE
    if overlay:
        code = code.replace("cpu = oracle_server(w, h); s.populate(cpu); cpu.render()",
                            "boxes = [((20.0, 30.0), (200.0, 90.0), (0.9, 0.2, 0.1, 0.5)), ((100.0, 60.0), (150.0, 120.0), (0.1, 0.8, 0.3, 0.35))]\n"
                            "cpu = oracle_server(w, h); s.populate(cpu); cpu.set_uiboxes(boxes); cpu.render()")
        code = code.replace("s.populate(vs)\nshots, notices = [], 0", "s.populate(vs)\nvs.set_uiboxes(boxes)\nshots, notices = [], 0")
        code = code.replace("s.populate(vs2)\nvs2.render()", "s.populate(vs2)\nvs2.set_uiboxes(boxes)\nvs2.render()")
        assert code.count("set_uiboxes") == 3
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env={**os.environ, "RND_TEST_SMALL_STREAMS": "1", "PYTHONPATH": root},
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "overflow ok" in out.stdout, (out.stdout[-800:], out.stderr[-2000:])
