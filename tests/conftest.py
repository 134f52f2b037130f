import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def orc():
    from oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def sc():
    from nrc_hpm_renderer_amd import scene
    return scene


@pytest.fixture(scope="session")
def cloud16():
    """the reference's data/volume/wdas_cloud_sixteenth.vdb, dense-ified and quantised (tests/golden/make_golden.py)"""
    return np.load(os.path.join(GOLDEN, "cloud_sixteenth_u8.npz"))["density"]


@pytest.fixture(scope="session")
def exr_stats():
    with open(os.path.join(GOLDEN, "exr_stats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def sphere_scene(sc):
    return sc.make_scene(sc.quantize_density(sc.sphere_volume(64)), scene_id=4)


@pytest.fixture(scope="session")
def api():
    from nrc_hpm_renderer_amd import api as a
    return a


@pytest.fixture(scope="session")
def torch_gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.cuda.set_device(0)
    return torch


FRAME_RANDOM = [0.25, 0.5, 0.75, 1.0]


def nrc_debug(monkeypatch, **switches):
    """NRC_DEBUG="name[=value],..." (csrc/nrc_common.hpp: the library's one environment switch, read when a cache / renderer is created):
    nrc_debug(monkeypatch, single_stream=True, wgrad_old=0); no arguments (or only False / None values) = unset"""
    parts = [k if v is True else "%s=%d" % (k, int(v)) for k, v in switches.items() if v is not None and v is not False]
    if parts:
        monkeypatch.setenv("NRC_DEBUG", ",".join(parts))
    else:
        monkeypatch.delenv("NRC_DEBUG", raising=False)
