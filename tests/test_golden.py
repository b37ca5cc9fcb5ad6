"""Committed regression vectors (tests/golden/, written by tests/golden/make_golden.py from the oracle).

They do not pin the oracle to the reference (which holds no fixtures for this path);
they freeze today's coverage / winning triangle / depth bits / RGBA8 so that drift shows up as a diff
against a committed file: the oracle must reproduce them exactly, the HIP path bit-exactly for the
integer planes and within the colour tolerance.
"""
import importlib.util
import os

import numpy as np
import pytest

from tests.helpers import ROOT, oracle_server

_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
make_golden = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(make_golden)

NAMES = sorted(make_golden.SCENES)


def load(name):
    with np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(name):
    want = load(name)
    s = make_golden.SCENES[name]()
    got = make_golden.render(s, oracle_server(s.width, s.height, threads=2))
    assert (got["ids"] == want["ids"]).all()
    assert (got["depth_bits"] == want["depth_bits"]).all()
    assert np.abs(got["rgba8"].astype(int) - want["rgba8"].astype(int)).max() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_path_matches_golden(name):
    from renderer_amd import VisualServer
    want = load(name)
    s = make_golden.SCENES[name]()
    got = make_golden.render(s, VisualServer(s.width, s.height, device=0))
    assert (got["ids"] == want["ids"]).all(), (got["ids"] != want["ids"]).sum()
    assert (got["depth_bits"] == want["depth_bits"]).all()
    diff = np.abs(got["rgba8"].astype(int) - want["rgba8"].astype(int)).max(axis=-1)
    assert (diff > 1).mean() <= 1e-3, (diff > 1).sum()      # shadow-tap flips only
