"""RCCL on the one GPU a builder can lease: `init_process_group("nccl", world_size=1)` in a FRESH child process with the
data-parallel exchange path forced (tests/workers/rccl_world1_worker.py).  What it pins: RCCL initialises and runs collectives on
this image; every collective call of the N > 1 PPO step (dtype views, storage offsets, async handles, issue order under both stream
schedules) is accepted by RCCL; the exchange path is bit-identical to the plain step when the collectives are identities.  What it
cannot pin: more than one rank (the driver's multi-GPU node)."""
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def test_rccl_world1_forced_exchange_is_bit_identical_to_the_plain_step(dev):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29633", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(REPO, "tests", "workers", "rccl_world1_worker.py")], env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "RCCL_WORLD1_OK" in out.stdout, out.stdout[-3000:] + out.stderr[-4000:]
    assert out.stdout.count("RCCL_WORLD1_EXCHANGE_BITEQUAL") == 4


def test_graphed_step_captures_its_rccl_collectives(dev):
    """ppo.GraphedPPOStep under data parallelism (round 4): rollout + update WITH the factor all-gathers, tail all-reduces, the
    RankLoss statistics and metric all-reduces captured in one HIP graph -- one RCCL rank, exchange forced, three steps (eager,
    capture + replay, replay; the schedulers move) bit-identical to the eager data-parallel steps.  A fresh child with its own short
    time limit: a capture that RCCL cannot serve must fail here, not hang the suite."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(REPO, "tests", "workers", "rccl_graph_worker.py")], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "RCCL_GRAPH_BITEQUAL" in out.stdout, out.stdout[-3000:] + out.stderr[-4000:]
