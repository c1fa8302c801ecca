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


def test_bench_gpus_2_starts_two_ranks_and_reports_them(dev):
    """`python bench.py --gpus 2` with no launcher in the environment (the driver's scaling command): the parent starts two ranks
    through torch.distributed.run, both run the head-only and the composed (`value`) loops with the data-parallel exchange active,
    and the relayed line reports n_gpus = 2 and two ranks seen by the all-reduce probe.  Two ranks share this box's one GPU, so the
    backend is gloo (LR2_BENCH_BACKEND: RCCL refuses two ranks on one device); one-stream schedule, as world > 1 defaults to."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(LR2_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras",
                        "--no-cpu-baseline", "--no-graph", "--no-profile"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks_seen"] == 2 and d["backend"] == "gloo" and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and d["head_only_steps_per_sec"] > 0


def test_two_rank_encoder_finetune_keeps_replicas_identical(dev):
    """finetune_ppo_step + finetune_pointwise_step on two gloo ranks sharing the GPU, a different batch per rank: the averaged encoder /
    embedding / head gradients leave both replicas with the same bits, and they moved (tests/workers/dp_finetune_worker.py)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29613", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29613", os.path.join(REPO, "tests", "workers", "dp_finetune_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "DP_FINETUNE_REPLICAS_IDENTICAL_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_two_rank_trad_pointwise_steps_match_the_whole_batch_step(dev):
    """pointwise_trad / pointwise_2data_trad on two gloo ranks sharing the GPU (the script the reference runs under DDP): gradients
    averaged over the ranks -> replicas bit-identical, and equal to one step on the concatenated batch up to summation order
    (tests/workers/dp_trad_worker.py)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29614", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29614", os.path.join(REPO, "tests", "workers", "dp_trad_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "DP_TRAD_REPLICAS_IDENTICAL_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
