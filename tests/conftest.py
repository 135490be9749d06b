import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def bunny():
    z = np.load(os.path.join(ROOT, "tests", "golden", "bunny.npz"))
    return {"bun0": z["bun0"], "bun4": z["bun4"]}


def make_context(device=0):
    """A pcl_amd.Context for the `-m gpu` modules.  PCLHIP_TEST_OPTIONS="lane_search=1,..." (read HERE, by the tests -- the
    library itself reads no environment variable) sets context options on it, so the whole tier can be run once more with,
    e.g., the per-lane seeded search switched on (profiles/r05_lane_search_fullsize_and_fuzz.txt)."""
    import pcl_amd
    ctx = pcl_amd.Context(device)
    for kv in [o for o in os.environ.get("PCLHIP_TEST_OPTIONS", "").split(",") if o]:
        name, _, value = kv.partition("=")
        ctx.setOption(name, float(value))
    return ctx
