"""CPU-only checks of the host side: state_dict contract, flag surface, schedule, NDCG, C-ABI export list,
"no fallback" behaviour, and the product/oracle separation.  No kernel is launched here."""
import argparse
import ctypes
import json
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from conftest import GOLD, REPO, load_golden

ARGS = dict(mode="reg", labels_num=3, seq_length=196, max_imgs=16, visual_feat_dim=768)


@pytest.fixture(scope="module")
def native():
    from lr2ppo_amd import _native
    _native.build()
    return _native


def test_state_dict_keys_match_reference_checkpoints():
    from lr2ppo_amd.finetune import ppo
    with open(os.path.join(GOLD, "keys.json")) as f:
        keys = json.load(f)
    # max_imgs=1 keeps the allocation small; only out_layer.fc1's input width depends on it
    small = argparse.Namespace(**{**ARGS, "max_imgs": 1})
    for kind, cls in (("actor", ppo.Actor), ("critic", ppo.Critic), ("reward", ppo.Reward)):
        mod = cls(small, None)
        got = [(n, list(p.shape)) for n, p in mod.named_parameters()]
        want = [(n, s if n != "out_layer.fc1.weight" else [3072, 197 * 768]) for n, s in keys[kind]]
        assert got == want, kind
        assert list(mod.state_dict().keys()) == [n for n, _ in keys[kind]]
    ac = ppo.ActorCritic(small, None)
    assert sorted({k.split(".")[0] for k in ac.state_dict()}) == keys["actor_critic_prefixes"]


def test_decay_groups_follow_the_substring_rule():
    from lr2ppo_amd.finetune import ppo
    small = argparse.Namespace(**{**ARGS, "max_imgs": 1})
    groups = ppo._grouped(list(ppo.Critic(small, None).named_parameters()))
    named = dict(ppo.Critic(small, None).named_parameters())
    n_decay = len(groups[0]["params"])
    n_nodecay = len(groups[1]["params"])
    names = list(named)
    assert n_nodecay == sum(1 for n in names if "bias" in n)          # no gamma/beta names in the head
    assert n_decay == len(names) - n_nodecay
    # LayerNorm *weights* and pos_emb are decayed
    decayed = {id(p) for p in groups[0]["params"]}
    c = ppo.Critic(small, None)
    g2 = ppo._grouped(list(c.named_parameters()))
    d2 = {id(p) for p in g2[0]["params"]}
    assert id(dict(c.named_parameters())["xit.1.0.weight"]) in d2 and id(dict(c.named_parameters())["pos_emb.weight"]) in d2
    assert decayed is not None


def test_linear_schedule_matches_reference_table():
    from lr2ppo_amd.tencentpretrain.utils.optimizers import get_linear_schedule_with_warmup
    with open(os.path.join(GOLD, "sched.json")) as f:
        s = json.load(f)
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=s["base_lr"])
    sch = get_linear_schedule_with_warmup(opt, s["warmup_steps"], s["train_steps"])
    lrs = [opt.param_groups[0]["lr"]]
    for _ in range(60):
        opt.step()
        sch.step()
        lrs.append(opt.param_groups[0]["lr"])
    assert np.allclose(lrs, s["lrs"], atol=1e-15)
    assert lrs[0] == 0.0


def test_ndcg_meter_matches_reference():
    from lr2ppo_amd.ndcg import AverageNDCGMeter
    g = load_golden("ndcg.npz")
    m = AverageNDCGMeter()
    for i in range(int(g["n_cases"])):
        got = m.return_ndcg_at_k_from_scores(g[f"scores_{i}"], g[f"gold_{i}"])
        assert torch.allclose(got, g[f"ndcg_{i}"], atol=1e-6), i
        m.compute_ndcg_at_k(g[f"gold_{i}"][torch.sort(g[f"scores_{i}"], descending=True)[1]],
                            torch.sort(g[f"gold_{i}"], descending=True)[0])
    vals = m.value()
    want = torch.stack([g[f"ndcg_{i}"] for i in range(int(g["n_cases"]))]).mean(0)
    assert torch.allclose(torch.stack([vals[k] for k in m.ndcg_at_k]), want, atol=1e-6)


def test_cli_accepts_the_reference_launcher_arguments(tmp_path):
    """Argument list of ppo.sh:42-63 (paths shortened) parses, and config < CLI precedence holds."""
    from lr2ppo_amd.finetune import ppo
    from lr2ppo_amd.tencentpretrain.utils.config import load_hyperparam
    cfg = tmp_path / "cfg.json"
    cfg.write_text(json.dumps({"emb_size": 768, "dropout": 0.3, "batch_size": 5, "seq_length": 11}))
    argv = ["--pretrained_model_path", "a.bin", "--reward_model_path", "r.bin", "--vit_pretrained_model_path", "v.bin",
            "--vocab_path", "vocab.json", "--merges_path", "merges.txt", "--tokenizer", "bpe", "--vit_tokenizer", "virtual",
            "--config_path", str(cfg), "--vit_config_path", str(cfg), "--train_path", "t.json", "--dev_path", "d.json",
            "--test_path", "x.json", "--output_model_path", "o.bin", "--epochs_num", "30", "--batch_size", "24",
            "--seq_length", "196", "--max_imgs", "16", "--visual_feat_dim", "768", "--mode", "reg", "--max_tags", "80",
            "--learning_rate", "1e-3", "--critic_learning_rate", "1e-3", "--max_timesteps", "1", "--update_timesteps", "200",
            "--eps_clip", "0.2", "--kl_div_loss_weight", "0.001", "--entropy_weight", "0.001", "--value_clip", "0.5",
            "--exp_name", "ppo", "--use_pairwise"]
    args = ppo.build_parser().parse_args(argv)
    assert args.update_timesteps == 200 and args.kl_div_loss_weight == 0.001 and args.seed == 7
    merged = load_hyperparam(args, argv=["prog"] + argv)
    assert merged.batch_size == 24 and merged.seq_length == 196      # CLI beats config
    assert merged.dropout == 0.3 and merged.emb_size == 768           # config beats defaults


def test_header_symbols_are_declared_bound_and_exported(native):
    hdr = open(os.path.join(REPO, "include", "lr2ppo_hip.h")).read()
    declared = set(re.findall(r"^int (lr2_[a-z0-9_]+)\(", hdr, flags=re.M))
    assert declared == set(native.SIGNATURES), declared ^ set(native.SIGNATURES)
    out = subprocess.run(["nm", "-D", "--defined-only", native.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (lr2_[a-z0-9_]+)", out))
    assert declared <= exported, declared - exported
    lib = native.lib()
    assert lib.lr2_abi_version() == 1
    assert ctypes.sizeof(native.Epilogue) == 5 * 8 + 4 * 4 + 2 * 4 + 4 + 4 + 4 + 4 + 8
    assert ctypes.sizeof(native.AdamChunk) == 48


def test_no_cpu_fallback_and_bad_arguments_are_errors(native):
    from lr2ppo_amd.finetune import ppo
    small = argparse.Namespace(**{**ARGS, "max_imgs": 1})
    actor = ppo.Actor(small, None)
    with pytest.raises(TypeError):
        actor(torch.zeros(1, 2, 196, 768), torch.zeros(1, 2, 1, 768), None)     # CPU tensors: refuse, don't emulate
    with pytest.raises(NotImplementedError):
        ppo.Actor(argparse.Namespace(**{**ARGS, "mode": "cls"}), None)
    with pytest.raises(ValueError):
        ppo.Actor(argparse.Namespace(**{**ARGS, "seq_length": 128}), None)
    # the C entry points validate before launching anything (no GPU is touched by a rejected call)
    lib = native.lib()
    assert lib.lr2_gemm(None, None, 1, 128, 64, 64, 64, 0, 0, 0, 0, None, None, 1, 128, 3, None) == -1
    assert lib.lr2_layernorm_fwd(None, None, None, None, None, None, 1, 768, 1e-5, 0, 0, 0, None) == -1
    assert lib.lr2_adamw_multi(None, 0, 1e-3, 0.9, 0.999, 1e-6, None) == -1


def test_product_never_imports_the_oracle_or_the_reference():
    bad = []
    for root, _, files in os.walk(os.path.join(REPO, "lr2ppo_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(root, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M) or "/root/reference" in src:
                    bad.append(f)
    assert not bad, bad


def test_synthetic_dataset_shapes_and_determinism():
    from lr2ppo_amd.finetune.ppo import SyntheticMovieNet
    ds = SyntheticMovieNet(4, 2, 16, seed=7)
    t, i, y = ds[1]
    assert t.shape == (2, 196, 768) and i.shape == (16, 768) and y.shape == (2,) and int(y.max()) <= 2
    t2, _, _ = SyntheticMovieNet(4, 2, 16, seed=7)[1]
    assert torch.equal(t, t2)
