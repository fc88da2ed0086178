import os
import sys

import numpy as np
import pytest

try:            # torch bundles its own HIP runtime: it must be the FIRST one loaded in a process that uses both
    import torch  # noqa: F401  (see wenet_amd/lib.py)
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
GOLDEN_CASES = ["v1_20dB", "v1_8dB", "v1_6dB", "v2_20dB", "v2_8dB", "v2_6dB", "v2_cs16_10dB", "v2_ppm150_12dB", "4fsk_12dB"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def have_gpu():
    try:
        from wenet_amd import lib
        return lib.load().wenet_rx_device_info(0) > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib as ol
    return ol.oracle()


@pytest.fixture(scope="session")
def ol():
    import oracle_lib
    oracle_lib.oracle()
    return oracle_lib


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(params=GOLDEN_CASES)
def golden(request):
    g = load_golden(request.param)
    g["name"] = request.param
    return g


def bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and bool((a.view(np.uint8) == b.view(np.uint8)).all())
