import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLD = os.path.join(REPO, "tests", "golden")

# one torch thread per USABLE core (cgroup quota): the CPU oracle is the checker of most GPU tests, and an oversubscribed quota makes
# it 2.2 x slower (oracle/cpu_threads.py)
from oracle.cpu_threads import fit_torch_threads  # noqa: E402

fit_torch_threads()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full-size CPU oracle checks (tens of seconds)")


def load_golden(name):
    z = np.load(os.path.join(GOLD, name), allow_pickle=False)
    return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def has_gpu():
    return torch.cuda.is_available()


@pytest.fixture(scope="session")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
