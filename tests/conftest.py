import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the oracle's OpenMP loop runs over envs; the tests use tens of envs, where more than a few threads only add spin-wait time
os.environ.setdefault("OMP_NUM_THREADS", "2")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "kick_env_golden.npz"))


@pytest.fixture(scope="session")
def model():
    import json
    return json.load(open(os.path.join(ROOT, "bez_isaacgym_amd", "model", "bez_model.json")))


def free_port():
    """A TCP port that is free on 127.0.0.1 right now (rendezvous of the subprocess tests: a fixed number may be taken on a shared box)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]
