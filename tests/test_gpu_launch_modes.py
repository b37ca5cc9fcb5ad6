"""Launch modes of round 3: the shadow passes on a stream of their own beside the main pass's geometry, and the measured
choice between overlapped and one-stream frames (windows of frames in each mode, a switch in between).  In both, a run of
DIFFERENT frames is enqueued without any wait, each frame into its own output buffer; afterwards every buffer must hold
exactly what the same frame gives when rendered alone and waited for.  A race between a frame's shadow pass and its
neighbours' shading or geometry, or a frame lost at a mode switch, shows as a differing buffer."""
import os
import subprocess
import sys

import numpy as np
import pytest

from renderer_amd import VisualServer
from renderer_amd.scenes import c1_sphere
from renderer_amd.types import Color, Light, from_translation, from_rotation_y, inverse, look_to_lh
from tests.helpers import oracle_server

pytestmark = pytest.mark.gpu


def sun(k):
    """A directional light whose direction changes from frame to frame: every frame has its own shadow maps."""
    t = inverse(look_to_lh((-1.5 + 0.4 * (k % 5), 20.0, -6.0 + 0.3 * (k % 3)), (0.05 + 0.03 * (k % 5), -1.0, 0.2 - 0.04 * (k % 4))))
    return t, Light.directional().with_color(Color(1.0, 0.9, 0.8, 3.5))


def run_frames(n_frames, width=320, height=192):
    import torch
    s = c1_sphere(width, height)
    vs = VisualServer(width, height, device=0)
    s.populate(vs)
    nbytes = width * height * 4
    # the frame buffer of a single-rank context is internal: copy it out stream-ordered after every frame
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    stream = torch.cuda.Stream()
    vs.set_stream(stream.cuda_stream)
    ptr, n = vs.color_device_ptr()
    assert n == nbytes
    bufs = [torch.zeros(nbytes, dtype=torch.uint8, device="cuda") for _ in range(n_frames)]
    torch.cuda.synchronize()
    cams = [from_translation((0.25 * (k % 3) - 0.25, 0.1 * (k % 2), -3.0 - 0.1 * (k % 4))) @ from_rotation_y(0.02 * (k % 7)) for k in range(n_frames)]
    with torch.cuda.stream(stream):
        for k in range(n_frames):
            vs.set_camera(cams[k], s.camera)
            vs.set_light(100, *sun(k))
            vs.render()
            assert hip.hipMemcpyAsync(C.c_void_p(bufs[k].data_ptr()), C.c_void_p(ptr), C.c_size_t(nbytes), 3, C.c_void_p(stream.cuda_stream)) == 0
    vs.finish()
    torch.cuda.synchronize()
    return s, cams, [b.cpu().numpy().reshape(height, width, 4) for b in bufs]


def check_against_alone(s, cams, shots, which):
    alone = VisualServer(s.width, s.height, device=0)
    s.populate(alone)
    distinct = set()
    for k in which:
        alone.set_camera(cams[k], s.camera)
        alone.set_light(100, *sun(k))
        alone.render()
        want = alone.read_color_rgba8()
        assert (want == shots[k]).all(), f"frame {k} differs from the same frame rendered alone on {(want != shots[k]).any(axis=-1).sum()} pixels"
        distinct.add(int(want.astype(np.int64).sum()))
    assert len(distinct) >= min(len(which), 6) - 1          # the frames really differ from each other
    return alone


def test_shadow_passes_beside_the_main_pass():
    s, cams, shots = run_frames(24)
    alone = check_against_alone(s, cams, shots, range(24))
    # and the last frame against the oracle (shadowed pixels included: the sphere shadows itself)
    cpu = oracle_server(s.width, s.height)
    s.populate(cpu)
    cpu.set_camera(cams[23], s.camera)
    cpu.set_light(100, *sun(23))
    cpu.render()
    assert (alone.read_triangle_ids() == cpu.read_triangle_ids()).all()
    d = np.abs(alone.read_color_rgba8().astype(int) - cpu.read_color_rgba8().astype(int))
    assert d.max() <= 1


def test_measured_launch_mode_switches_keep_every_frame():
    """More frames than the two measurement windows and the decision take (8 + 24 frames per mode), no wait in between:
    the mode changes at least twice inside the run."""
    s, cams, shots = run_frames(100)
    check_against_alone(s, cams, shots, list(range(0, 100, 7)) + [30, 31, 32, 33, 34, 63, 64, 65, 66, 67, 99])


@pytest.mark.parametrize("env", [{"RND_SHADOW_STREAM": "0"}, {"RND_TUNE": "0"}, {"RND_SHADOW_STREAM": "0", "RND_OVERLAP": "0"}])
def test_modes_switched_off(env):
    """The same run with the shadow stream, the measurement or both overlaps switched off at context creation (child
    process: the switches are read there)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("from tests.test_gpu_launch_modes import run_frames, check_against_alone\n"
            "s, cams, shots = run_frames(40)\ncheck_against_alone(s, cams, shots, range(0, 40, 3))\nprint('modes ok')\n")
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env={**os.environ, **env, "PYTHONPATH": root},
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "modes ok" in out.stdout, (out.stdout[-800:], out.stderr[-2000:])
