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
