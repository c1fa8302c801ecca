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


# The big seeded parameter sets (0.5 B normals from ONE CPU generator: ~3 s each) are asked for by dozens of tests with the same
# (spec, seed): keep the few distinct ones and hand out copies (callers scale tensors in place).  Test infrastructure only.
from collections import OrderedDict  # noqa: E402

from oracle import lr2ppo_oracle as _O  # noqa: E402

_seeded_params_raw = _O.seeded_params
_SEEDED_CACHE: "OrderedDict" = OrderedDict()


def _seeded_params_cached(spec, seed, std=0.02, skip_gamma_beta=True):
    spec = list(spec)
    numel = sum(int(np.prod(shape)) for _, shape in spec)
    if numel < 50_000_000:
        return _seeded_params_raw(spec, seed, std, skip_gamma_beta)
    key = (tuple((n, tuple(sh)) for n, sh in spec), int(seed), float(std), bool(skip_gamma_beta))
    hit = _SEEDED_CACHE.get(key)
    if hit is None:
        hit = _seeded_params_raw(spec, seed, std, skip_gamma_beta)
        _SEEDED_CACHE[key] = hit
        while len(_SEEDED_CACHE) > 6:                       # <= 6 x 2 GB of host memory
            _SEEDED_CACHE.popitem(last=False)
    else:
        _SEEDED_CACHE.move_to_end(key)
    return {k: v.clone() for k, v in hit.items()}


_O.seeded_params = _seeded_params_cached


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
