"""BASELINE configs[0] and the `_trad` transfer experiment END TO END on tiny synthetic LETOR files, through the entry points only
(finetune/*_trad.sh upstream): raw TSVs (46- and 136-wide rows) -> per-query HDF5 -> pointwise_2data_trad (the 46 / 136 -> 768
projections) -> pointwise_2data_infer_trad (projected TSVs) -> HDF5 -> pointwise_trad (stage 1) and reward_trad (stage 2) ->
ppo_trad (stage 3, actor from stage 1, critic / reward from stage 2, strict loads) -> ppo_eval_trad.  Every hand-over is a file the
next script reads the way the reference's script would."""
import csv
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu

FLAGS = ["--config_path", "lr2ppo_amd/configs/roberta_base.json", "--vit_config_path", "lr2ppo_amd/configs/vit_base_16_224.json",
         "--seq_length", "196", "--max_imgs", "16", "--visual_feat_dim", "768", "--learning_rate", "1e-4", "--mode", "reg"]


def _run(module_or_script, *argv, port):
    head = [sys.executable, module_or_script] if module_or_script.endswith(".py") else [sys.executable, "-m", module_or_script]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PYTHONPATH=REPO)
    r = subprocess.run(head + list(argv), cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (module_or_script, r.stdout[-1500:], r.stderr[-3000:])
    return r.stdout


def _tsv(path, rng, width, queries):
    rows = []
    for qid, n in queries:
        for _ in range(n):
            rows.append([int(rng.randint(0, 3)), qid] + [round(float(v), 5) for v in rng.standard_normal(width)])
    with open(path, "w") as f:
        csv.writer(f, delimiter="\t").writerows(rows)
    return len(rows)


def test_trad_experiment_runs_from_tsv_files_to_the_stage3_evaluation(dev, tmp_path):
    try:
        from lr2ppo_amd import h5lite
        h5lite.library()
    except ImportError:
        pytest.skip("no HDF5 C library on this host")
    rng = np.random.RandomState(11)
    raw = {name: tmp_path / f"raw_{name}" for name in ("mq2008", "web10k")}
    n_rows = {}
    for name, width in (("mq2008", 46), ("web10k", 136)):
        raw[name].mkdir()
        for split, queries in (("train", [(10 + i, 14 + 3 * i) for i in range(6)]), ("test", [(50 + i, 25 - 2 * i) for i in range(4)])):
            n_rows[name, split] = _tsv(raw[name] / f"{split}.tsv", rng, width, queries)
    conv = os.path.join(REPO, "tools", "convert_to_h5py.py")
    h5 = {name: str(tmp_path / f"h5_{name}") for name in raw}
    for name in raw:
        _run(conv, "--original_dir", str(raw[name]), "--target_dir", h5[name], port=0)
    # stage 0: the projections (BASELINE configs[0]: the 46- / 136-dim MLP ranker)
    proj = str(tmp_path / "proj.bin")
    _run("lr2ppo_amd.finetune.pointwise_2data_trad", *FLAGS, "--train_path", h5["mq2008"], "--train_path2", h5["web10k"], "--dev_path", h5["mq2008"],
         "--batch_size", "2", "--epochs_num", "1", "--report_steps", "2", "--output_model_path", proj, "--log_path", str(tmp_path / "s0.log"), port=29701)
    assert "text_proj3.fc2.weight" in torch.load(proj, map_location="cpu")
    # the dimension projection: 46-wide MQ2008 rows -> 768 features, same rows, same leading columns
    tsv768 = tmp_path / "tsv768"
    out = _run("lr2ppo_amd.finetune.pointwise_2data_infer_trad", *FLAGS, "--train_path", "x", "--dev_path", "x", "--dim_proj_ckpt_path", proj,
               "--input_dir", str(raw["mq2008"]), "--output_dir", str(tsv768), port=0)
    assert f"train.tsv: {n_rows['mq2008', 'train']} rows" in out
    with open(tsv768 / "test.tsv") as f:
        first = next(csv.reader(f, delimiter="\t"))
    assert len(first) == 2 + 768
    feats = str(tmp_path / "h5_768")
    _run(conv, "--original_dir", str(tsv768), "--target_dir", feats, port=0)
    with h5lite.File(os.path.join(feats, "train.h5")) as f:
        assert len(f) == 6 and f[f.keys()[0]].shape == (20, 770)
    # stage 1 and stage 2 on the projected features
    stage1, stage2 = str(tmp_path / "stage1.bin"), str(tmp_path / "stage2.bin")
    _run("lr2ppo_amd.finetune.pointwise_trad", *FLAGS, "--train_path", feats, "--dev_path", feats, "--batch_size", "2", "--epochs_num", "1",
         "--report_steps", "2", "--output_model_path", stage1, "--log_path", str(tmp_path / "s1.log"), port=29702)
    _run("lr2ppo_amd.finetune.reward_trad", *FLAGS, "--train_path", feats, "--dev_path", feats, "--batch_size", "8", "--epochs_num", "1",
         "--report_steps", "2", "--output_model_path", stage2, "--log_path", str(tmp_path / "s2.log"), port=29703)
    assert "val accuracy:" in open(tmp_path / "s2.log").read()
    # stage 3 from both checkpoints (strict loads: actor <- stage 1, critic and reward <- stage 2), then the evaluation-only script
    stage3 = str(tmp_path / "stage3.bin")
    s3 = ["--train_path", feats, "--dev_path", feats, "--batch_size", "4", "--epochs_num", "2", "--critic_learning_rate", "1e-4",
          "--max_timesteps", "1", "--update_timesteps", "2", "--kl_div_loss_weight", "0.001", "--entropy_weight", "0.001", "--value_clip", "0.5",
          "--max_cycles", "1"]
    _run("lr2ppo_amd.finetune.ppo_trad", *FLAGS, *s3, "--pretrained_model_path", stage1, "--reward_model_path", stage2,
         "--output_model_path", stage3, "--log_path", str(tmp_path / "s3.log"), port=29704)
    trained = [ln for ln in open(tmp_path / "s3.log").read().splitlines() if ln.startswith("NDCG@3=")][-1]
    _run("lr2ppo_amd.finetune.ppo_eval_trad", *FLAGS, *s3, "--pretrained_model_path", stage3, "--output_model_path", str(tmp_path / "unused.bin"),
         "--log_path", str(tmp_path / "s4.log"), port=29705)
    assert [ln for ln in open(tmp_path / "s4.log").read().splitlines() if ln.startswith("NDCG@3=")][-1] == trained
