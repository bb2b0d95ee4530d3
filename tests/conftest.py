import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN, "golden.json")) as f:
        g = json.load(f)
    g["dir"] = GOLDEN
    return g


@pytest.fixture(scope="session")
def oracle_built():
    """libsw_oracle.so (and oracle/_ref when /root/reference exists) -- building the checker."""
    so = os.path.join(ROOT, "oracle", "libsw_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return so


def load_npy(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
