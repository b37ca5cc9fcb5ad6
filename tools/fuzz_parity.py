"""Randomized parity sweep (not part of the test suite): random triangle soups, sizes, settings and 2D
content through the HIP path and the CPU oracle; prints one line per case and a summary.
    python tools/fuzz_parity.py [n_cases] [first_seed]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import VisualServer  # noqa: E402
from tests.helpers import oracle_server  # noqa: E402
from tests.test_gpu_parity import overlay_content, soup_scene  # noqa: E402


def run_case(seed):
    """One random scene through the HIP path and the CPU oracle; returns (description, ids equal, depth bits equal,
    largest colour difference, fraction of pixels above 1 LSB)."""
    r = np.random.default_rng(seed)
    w, h = int(r.integers(17, 500)), int(r.integers(17, 400))
    size = float(r.choice([0.02, 0.1, 0.6, 3.0]))
    s = soup_scene(seed, w, h, n_meshes=int(r.integers(1, 6)), n_tri=int(r.integers(1, 400)), size=size,
                   lights=int(r.integers(0, 6)), depth=(float(r.choice([-1.0, 0.3])), 8.0))
    if r.random() < 0.5:
        s.lights = [l for l in s.lights if l[2].kind_id() == 1]
    msaa = 4 if r.random() < 0.25 else 1
    factor = float(r.choice([1.0, 1.0, 0.5, 1.5]))
    tone = int(r.integers(0, 2))
    ui = r.random() < 0.4
    pair = []
    for make in (lambda: VisualServer(w, h, device=0), lambda: oracle_server(w, h)):
        vs = make()
        vs.set_tone_mapping(tone)
        vs.set_msaa(msaa)
        vs.set_render_size_factor(factor)
        s.populate(vs)
        if ui:
            overlay_content(vs, seed, w, h)
        vs.render()
        pair.append(vs)
    g, c = pair
    ids_ok = bool((g.read_triangle_ids() == c.read_triangle_ids()).all())
    depth_ok = bool((g.read_depth().view(np.uint32) == c.read_depth().view(np.uint32)).all())
    d = np.abs(g.read_color_rgba8().astype(int) - c.read_color_rgba8().astype(int)).max(axis=-1)
    frac = float((d > 1).mean())
    g.close(); c.close()
    desc = f"{w}x{h} msaa {msaa} factor {factor} tone {tone} ui {int(ui)} lights {len(s.lights)} size {size}"
    return desc, ids_ok, depth_ok, int(d.max()), frac


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    bad = 0
    for seed in range(first, first + n):
        desc, ids_ok, depth_ok, dmax, frac = run_case(seed)
        ok = ids_ok and depth_ok and frac <= 1e-3
        bad += not ok
        print(f"seed {seed}: {desc}: ids {ids_ok} depth {depth_ok} colour max {dmax} >1LSB {frac:.2e} {'ok' if ok else 'MISMATCH'}", flush=True)
    print(f"{n - bad} of {n} cases within the bar")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
