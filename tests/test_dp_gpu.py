"""N > 1 rehearsal on the one-GPU box: two gloo ranks share cuda:0 and run the data-parallel PPO update (factor
all-gather + tail all-reduce + fused out_layer.fc1 update); RCCL itself needs the driver's multi-GPU node."""
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fuse", ["1", "0"])
def test_two_rank_update_keeps_replicas_identical(dev, fuse):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", LR2_TEST_FUSE=fuse, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29611", os.path.join(REPO, "tests", "workers", "dp_gpu_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "DP_REHEARSAL_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
    # the exchanged gradient equals the mean of the two ranks' independently computed gradients (distinct data per rank),
    # also through the fused out_layer.fc1 update; RankLoss statistics are global
    assert "DP_GRADIENT_IS_RANK_MEAN_OK" in out.stdout and "GLOBAL_RANK_LOSS_OK" in out.stdout, out.stdout[-2000:]
