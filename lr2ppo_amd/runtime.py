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


_dev_seed = None        # device-seed mode (capture of a training step): [int64[1] device tensor, draws so far]


class device_seed:
    """Context: every next_drop() inside draws `*tensor + i` (i = 0, 1, ... in call order) instead of the host counter, so the
    launches can be captured in a HIP graph; `.draws` is the number of seeds the region consumed.  Replaying the region with
    *tensor = peek_drop_seed() followed by advance(draws) gives the masks the eager calls would have had."""

    def __init__(self, tensor: torch.Tensor):
        if tensor.dtype != torch.int64 or tensor.numel() != 1 or not tensor.is_cuda:
            raise TypeError("device_seed: an int64[1] HIP tensor")
        self.state = [tensor, 0]

    @property
    def draws(self) -> int:
        return self.state[1]

    def __enter__(self):
        global _dev_seed
        if _dev_seed is not None:
            raise RuntimeError("device_seed regions do not nest")
        _dev_seed = self.state
        return self

    def __exit__(self, *exc):
        global _dev_seed
        _dev_seed = None
        return False


def advance(calls: int):
    """Consume `calls` seeds of the host stream (a replayed captured step drew them on the device)."""
    global _drop_seed, _drop_calls
    if _drop_seed is None:
        _drop_seed = int(torch.initial_seed()) & 0xFFFFFFFF
    _drop_calls += int(calls)


def next_drop(p: float, site_base: int, advance: bool = True) -> DropCfg:
    global _drop_seed, _drop_calls
    if _dev_seed is not None:
        i = _dev_seed[1]
        if advance:
            _dev_seed[1] += 1
        return DropCfg(p, i, site_base, seed_dev=_dev_seed[0])
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
