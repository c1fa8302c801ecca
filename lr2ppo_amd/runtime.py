"""Process-wide runtime state of the HIP path: dropout stream and data-parallel context."""
from __future__ import annotations

import torch

from .engine import DropCfg

_drop_seed = None
_drop_calls = 0


def set_dropout_seed(seed: int, calls: int = 0):
    """Seed of the counter-based dropout masks; every train-mode forward consumes one call index."""
    global _drop_seed, _drop_calls
    _drop_seed, _drop_calls = int(seed), int(calls)


def next_drop(p: float, site_base: int, advance: bool = True) -> DropCfg:
    global _drop_seed, _drop_calls
    if _drop_seed is None:
        _drop_seed = int(torch.initial_seed()) & 0xFFFFFFFF
    seed = (_drop_seed << 24) + _drop_calls
    if advance:
        _drop_calls += 1
    return DropCfg(p, seed, site_base)


def peek_drop_seed() -> int:
    """The seed the next train-mode forward will use (tests feed it to the oracle's mask function)."""
    s = _drop_seed if _drop_seed is not None else int(torch.initial_seed()) & 0xFFFFFFFF
    return (s << 24) + _drop_calls
