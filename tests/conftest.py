import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
GOLDEN_CASES = ["G1", "G2", "G3", "G4", "G5", "G6", "G7", "G8"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


def load_golden(name):
    """Load one fixture written by tests/golden/make_goldens.py (data only)."""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    g = {k: z[k] for k in z.files}
    if "loc" in g:
        g["loc"] = "GC" if str(g["loc"]) == "GC" else None
    g["name"] = name
    return g


@pytest.fixture(params=GOLDEN_CASES)
def golden(request):
    return load_golden(request.param)
