"""CPU-only checks of the host side: state_dict contract, flag surface, schedule, NDCG, C-ABI export list,
"no fallback" behaviour, and the product/oracle separation.  No kernel is launched here."""
import argparse
import ctypes
import json
import os
import re
import subprocess
import sys

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


@pytest.fixture(scope="module")
def small_models():
    """One ActorCritic and one Reward for the whole module (each head holds a 1.9 GB out_layer.fc1 even at
    max_imgs=1, the smallest the architecture allows -- only that layer's input width depends on max_imgs)."""
    from lr2ppo_amd.finetune import ppo
    small = argparse.Namespace(**{**ARGS, "max_imgs": 1})
    with torch.device("meta"):         # names, shapes, grouping and argument checks need no storage (3 x 2 GB and their init otherwise)
        ac = ppo.ActorCritic(small, None)
        reward = ppo.Reward(small, None)
    return {"actor": ac.actor, "critic": ac.critic, "reward": reward, "actor_critic": ac}


def test_state_dict_keys_match_reference_checkpoints(small_models):
    with open(os.path.join(GOLD, "keys.json")) as f:
        keys = json.load(f)
    for kind in ("actor", "critic", "reward"):
        mod = small_models[kind]
        got = [(n, list(p.shape)) for n, p in mod.named_parameters()]
        want = [(n, s if n != "out_layer.fc1.weight" else [3072, 197 * 768]) for n, s in keys[kind]]
        assert got == want, kind
        assert list(mod.state_dict().keys()) == [n for n, _ in keys[kind]]
    ac = small_models["actor_critic"]
    assert sorted({k.split(".")[0] for k in ac.state_dict()}) == keys["actor_critic_prefixes"]


def test_decay_groups_follow_the_substring_rule(small_models):
    from lr2ppo_amd.finetune import ppo
    c = small_models["critic"]
    named = dict(c.named_parameters())
    groups = ppo._grouped(list(named.items()))
    names = list(named)
    assert len(groups[1]["params"]) == sum(1 for n in names if "bias" in n)   # no gamma/beta names in the head
    assert len(groups[0]["params"]) == len(names) - len(groups[1]["params"])
    # LayerNorm *weights* and pos_emb are decayed
    decayed = {id(p) for p in groups[0]["params"]}
    assert id(named["xit.1.0.weight"]) in decayed and id(named["pos_emb.weight"]) in decayed


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
    assert lib.lr2_abi_version() == int(re.search(r"#define LR2_ABI_VERSION (\d+)", hdr).group(1)) == native.ABI_VERSION


def test_ctypes_structs_have_the_layout_the_c_compiler_gives_the_header(native, tmp_path):
    """sizeof / offsetof of every struct in include/lr2ppo_hip.h, as gcc lays it out, against the ctypes mirrors."""
    structs = {"lr2_epilogue": native.Epilogue, "lr2_adamw_chunk": native.AdamChunk, "lr2_split_chunk": native.SplitChunk}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "lr2ppo_hip.h"', 'int main(void) {']
    for cname, ct in structs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in ct._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for cname, ct in structs.items():
        assert int(got[cname]) == ctypes.sizeof(ct), cname
        for fname, _ in ct._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(ct, fname).offset, (cname, fname)


def test_no_cpu_fallback_and_bad_arguments_are_errors(native, small_models):
    from lr2ppo_amd.finetune import ppo
    actor = small_models["actor"]
    with pytest.raises(TypeError):
        actor(torch.zeros(1, 2, 196, 768), torch.zeros(1, 2, 1, 768), None)     # CPU tensors: refuse, don't emulate
    with torch.device("meta"):
        with pytest.raises(ValueError):
            ppo.Actor(argparse.Namespace(**{**ARGS, "mode": "rank"}), None)          # 'reg' and 'cls' only (ppo.py:209-212)
        assert ppo.Actor(argparse.Namespace(**{**ARGS, "mode": "cls"}), None).n_out == 3
        with pytest.raises(ValueError):
            ppo.Actor(argparse.Namespace(**{**ARGS, "seq_length": 128}), None)
    # the C entry points validate before launching anything (no GPU is touched by a rejected call)
    lib = native.lib()
    assert lib.lr2_gemm(None, None, 1, 128, 64, 64, 64, 0, 0, 0, 0, 0, 0, 0, 0, None, None, 1, 128, 3, None) == -1
    assert lib.lr2_layernorm_fwd(None, None, None, None, None, 0, None, None, 1, 768, 1e-5, 0, 0, 0, None) == -1
    assert lib.lr2_adamw_multi(None, 0, 1e-3, 0.9, 0.999, 1e-6, None, None) == -1
    assert lib.lr2_step_scalars_store(None, 1, None, 0, None) == -1
    assert lib.lr2_quant_mxfp8(None, 0, None, None, 1, 32, None) == -1
    assert lib.lr2_layernorm_fwd_mxfp8(None, None, None, None, None, None, 1, 768, 1e-6, 1, None) == -1
    assert lib.lr2_gemm_mxfp8(None, None, None, None, None, 0, None, None, 0, 0, None, None, None, 0, 0, 1, 128, 128, None) == -1
    assert lib.lr2_step_scalars_store(None, 1, None, 15, None) == -1         # at most LR2_STEP_SCALARS_MAX_LRS rates


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


def test_stage1_stage2_modules_mirror_the_reference_surface():
    """finetune/pointwise.py and finetune/reward_pair_dataloader.py: same public names; host-side helpers checked
    against fixtures generated from the imported reference (get_index) or worked out from pointwise.py:96-119."""
    import random
    from lr2ppo_amd.finetune import pointwise as pw, pointwise_trad as pt, reward_pair_dataloader as rp
    for n in ("Mlp", "Classifier", "load_or_initialize_parameters", "build_optimizer", "train_model"):
        assert hasattr(pt, n), n
    from lr2ppo_amd.finetune import ppo_trad
    from oracle import lr2ppo_oracle as O
    for n in ("RankLoss", "ActorCritic", "Actor", "Critic", "Reward", "load_or_initialize_parameters",
              "load_or_initialize_parameters_reward", "build_optimizer", "clipped_value_loss", "train_model", "evaluate"):
        assert hasattr(ppo_trad, n), n                                       # finetune/ppo_trad.py's public names
    from lr2ppo_amd.finetune import pointwise_2data_trad as p2, reward_trad as rt
    for mod, names in ((p2, ("Mlp", "Classifier", "load_or_initialize_parameters", "build_optimizer", "train_model")),
                       (rt, ("Classifier", "load_or_initialize_parameters", "build_optimizer", "train_model", "evaluate"))):
        for n in names:
            assert hasattr(mod, n), (mod.__name__, n)
    a = argparse.Namespace(mode="reg", labels_num=3)
    assert [(k, tuple(v.shape)) for k, v in p2.Classifier(a, None).state_dict().items()] == O.trad2_param_spec()
    assert [(k, tuple(v.shape)) for k, v in rt.Classifier(a, None).state_dict().items()] == O.trad_head_param_spec("reward")
    for cls_, kind in ((ppo_trad.Actor, "actor"), (ppo_trad.Critic, "critic"), (ppo_trad.Reward, "reward")):
        m = cls_(a, None)
        assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == O.trad_head_param_spec(kind)   # pinned by gen_ppo_trad
    for n in ("get_scores", "log_sig", "get_def_cls", "MovieNet", "Mlp", "Classifier", "load_or_initialize_parameters",
              "build_optimizer", "train_model", "evaluate", "get_dataloader", "main"):
        assert hasattr(pw, n), n
    for n in ("log_sig", "get_def_cls", "get_index", "MovieNet", "Mlp", "Classifier", "load_or_initialize_parameters",
              "build_optimizer", "train_model", "evaluate", "get_dataloader", "main"):
        assert hasattr(rp, n), n
    with open(os.path.join(GOLD, "stage2_get_index.json")) as f:
        cases = json.load(f)
    for seed, c in zip((1, 2, 3, 4, 5), cases):
        random.seed(seed)
        ch, rj = rp.get_index([{"target": t} for t in c["targets"]])
        assert ch == c["chosen"] and rj == c["reject"], c
    # training reader of stage 1: cut to max_tags, else pad by cycling through the non-zero-label tags (or all tags)
    assert pw.train_tag_index([0, 1, 0], 7) == [0, 1, 2, 1, 1, 1, 1]
    assert pw.train_tag_index([0, 2, 1, 0], 7) == [0, 1, 2, 3, 1, 2, 1]     # add_list = [1, 2]; i % 2 for i = 4, 5, 6
    assert pw.train_tag_index([0, 0], 5) == [0, 1, 0, 1, 0]
    assert pw.train_tag_index([1, 2, 0, 1], 3) == [0, 1, 2]
    # synthetic pairs follow the reader's layouts
    ds = rp.SyntheticPairs(6, True)
    for i in range(6):
        t, im, lab, ch, rj = ds[i]
        assert t.shape == (2, 196, 768) and (ch.tolist(), rj.tolist()) in [tuple(map(list, x)) for x in rp.TRAIN_LAYOUTS]
    t, im, lab, ch, rj = rp.SyntheticPairs(3, False)[1]
    assert t.shape == (3, 196, 768) and ch[:2].tolist() == rj[:2].tolist() and ch[2:].tolist() == rj[2:].tolist()[::-1]
    assert lab[ch[2]] >= lab[ch[3]]


def test_checkpoint_layouts_are_the_ones_that_crossed_to_the_reference_and_back():
    """SURVEY 8(f)-3, the `.bin` half.  oracle/gen_golden.py::gen_bin_interchange wrote each reference module with the reference's
    save_model, loaded it strict=True through the product's loader, wrote the product's checkpoint with the product's save_model and
    loaded THAT strict=True into the reference module -- every tensor equal both ways, for the 11 model classes -- and froze the layout
    that made the trip (names, shapes, dtypes in state_dict order).  Here: the product's modules still have exactly that layout
    (built on the meta device: no 2-GB allocations)."""
    import hashlib
    from lr2ppo_amd.finetune import (pointwise, pointwise_2data_trad, pointwise_trad, ppo, ppo_trad, reward_pair_dataloader,
                                     reward_trad)
    with open(os.path.join(GOLD, "bin_interchange.json")) as f:
        gold = json.load(f)
    classes = {"ppo.Actor": ppo.Actor, "ppo.Critic": ppo.Critic, "ppo.Reward": ppo.Reward, "pointwise.Classifier": pointwise.Classifier,
               "reward_pair_dataloader.Classifier": reward_pair_dataloader.Classifier, "ppo_trad.Actor": ppo_trad.Actor,
               "ppo_trad.Critic": ppo_trad.Critic, "ppo_trad.Reward": ppo_trad.Reward, "pointwise_trad.Classifier": pointwise_trad.Classifier,
               "pointwise_2data_trad.Classifier": pointwise_2data_trad.Classifier, "reward_trad.Classifier": reward_trad.Classifier}
    assert set(gold) == set(classes) | {"tower.vit", "tower.roberta"}
    from lr2ppo_amd.finetune.features import TEXT_CONFIG, VIT_CONFIG, EncoderStack, encoder_args
    for tower, cfg in (("tower.vit", VIT_CONFIG), ("tower.roberta", TEXT_CONFIG)):   # released ViT-B/16 / RoBERTa-base checkpoints' layout
        with torch.device("meta"):
            sd = EncoderStack(encoder_args(cfg), 50265).state_dict()
        text = ";".join(f"{k}:{tuple(v.shape)}:{str(v.dtype).replace('torch.', '')}" for k, v in sd.items())
        assert (len(sd), hashlib.sha256(text.encode()).hexdigest()) == (gold[tower]["tensors"], gold[tower]["layout_sha256"]), tower
    args = argparse.Namespace(mode="reg", labels_num=3, seq_length=196, max_imgs=16, visual_feat_dim=768)
    for name, cls in classes.items():
        with torch.device("meta"):
            sd = cls(args, None).state_dict()
        text = ";".join(f"{k}:{tuple(v.shape)}:{str(v.dtype).replace('torch.', '')}" for k, v in sd.items())
        g = gold[name]
        assert (len(sd), sum(v.numel() for v in sd.values())) == (g["tensors"], g["elements"]), name
        assert hashlib.sha256(text.encode()).hexdigest() == g["layout_sha256"], name
        assert g["reference_to_product"] == g["product_to_reference"] == "strict load, every tensor equal"


def test_stage_launcher_arguments_parse():
    """Argument lists of pointwise.sh and reward_pair_dataloader.sh (paths shortened)."""
    from lr2ppo_amd.finetune import pointwise as pw, reward_pair_dataloader as rp
    common = ["--train_path", "t.json", "--dev_path", "d.json", "--test_path", "x.json", "--epochs_num", "10",
              "--learning_rate", "1e-5", "--mask", "fully_visible", "--output_model_path", "o.bin", "--log_path", "l.txt",
              "--exp_name", "e", "--seq_length", "196", "--visual_feat_dim", "768", "--max_imgs", "16", "--max_tags", "20",
              "--pretrained_model_path", "r.bin", "--vocab_path", "v.txt", "--merges_path", "m.txt", "--tokenizer", "bpe",
              "--config_path", "c.json", "--encoder", "transformer", "--vit_pretrained_model_path", "vit.bin",
              "--vit_tokenizer", "virtual", "--vit_config_path", "vc.json", "--vit_encoder", "transformer"]
    a = pw.build_parser().parse_args(common + ["--batch_size", "2", "--report_steps", "150", "--mode", "reg"])
    assert a.batch_size == 2 and a.max_tags == 20 and a.mode == "reg"
    b = rp.build_parser().parse_args(common + ["--batch_size", "64", "--report_steps", "100", "--mode", "cls"])
    assert b.batch_size == 64 and b.mode == "cls"


def test_movienet_readers_match_reference_on_a_fake_h5(tmp_path, monkeypatch):
    """A13 / 8f-3: the three LRMovieNet readers (stage 1, 2, 3; train and validation splits) against the outputs of the
    reference's readers on the same in-memory stand-in for clean_feat.h5 (tests/golden/readers.json): tag selection and
    padding, pair layouts, get_index ordering, image shuffle + cyclic padding, labels -- with the RNGs seeded alike."""
    import random
    import sys
    import types
    from oracle import lr2ppo_oracle as O
    from lr2ppo_amd.finetune import pointwise as pw, ppo, reward_pair_dataloader as rp
    items, h5 = O.fake_movienet()
    fake = types.ModuleType("h5py")
    fake.File = lambda *a, **k: h5
    monkeypatch.setitem(sys.modules, "h5py", fake)
    path = tmp_path / "split.json"
    path.write_text(json.dumps(items))
    with open(os.path.join(GOLD, "readers.json")) as f:
        gold = json.load(f)
    for name, mod in (("ppo", ppo), ("pointwise", pw), ("reward_pair", rp)):
        for split in ("train", "val"):
            g = gold[f"{name}_{split}"]
            random.seed(11), np.random.seed(12), torch.manual_seed(13)
            ds = mod.MovieNet(argparse.Namespace(is_master=False, max_imgs=16, max_tags=g["max_tags"]), str(path),
                              is_train=split == "train")
            torch.manual_seed(14)
            assert len(ds) == g["len"], (name, split)
            for i, want in enumerate(g["items"]):
                got = O.describe_reader_item(ds[i])
                assert got == want, (name, split, i, got, want)


def test_dual_embedding_and_encoder_keys_match_reference_order():
    """dual_embedding.py:15-37 / dual_encoder.py:13-25: parameter names, shapes and registration order of the two-stream
    wrappers (incl. tie_weights) equal the reference's (tests/golden/dual_keys.json, frozen from the imported reference)."""
    import argparse
    from lr2ppo_amd.tencentpretrain.embeddings import DualEmbedding
    from lr2ppo_amd.tencentpretrain.encoders import DualEncoder
    from lr2ppo_amd.tencentpretrain.opts import finetune_opts, tokenizer_opts
    text = {"embedding": ["word", "pos", "seg"], "encoder": "transformer", "remove_embedding_layernorm": False,
            "layernorm_positioning": "post", "max_seq_length": 20, "layers_num": 1}
    vit = {"embedding": ["patch", "pos"], "encoder": "transformer", "remove_embedding_layernorm": True,
           "layernorm_positioning": "pre", "max_seq_length": 25, "layers_num": 1, "image_height": 32, "image_width": 48,
           "patch_size": 8, "channels_num": 3}
    keys = json.load(open(os.path.join(GOLD, "dual_keys.json")))
    for tag, s0, s1, tie in (("tv", text, vit, False), ("tt", text, text, True)):
        p = argparse.ArgumentParser()
        finetune_opts(p)
        tokenizer_opts(p)
        d = vars(p.parse_args([]))
        d.update(emb_size=128, hidden_size=128, feedforward_size=256, heads_num=2, layers_num=1, dropout=0.1, hidden_act="gelu",
                 embedding=["dual"], encoder="dual", stream_0=dict(s0), stream_1=dict(s1), tie_weights=tie, mask="fully_visible",
                 image_height=32, image_width=48, patch_size=8, channels_num=3)
        a = argparse.Namespace(**d)
        emb, enc = DualEmbedding(a, 100), DualEncoder(a)
        assert [[n, list(q.shape)] for n, q in emb.named_parameters()] == keys[tag]["embedding"], tag
        assert [[n, list(q.shape)] for n, q in enc.named_parameters()] == keys[tag]["encoder"], tag
        if tie:
            assert emb.embedding_0 is emb.embedding_1 and enc.encoder_0 is enc.encoder_1


def test_integration_doc_quotes_the_current_abi_version(native):
    doc = open(os.path.join(REPO, "INTEGRATION.md")).read()
    m = re.search(r"lr2_abi_version\(\) == (\d+)", doc)
    assert m and int(m.group(1)) == native.ABI_VERSION


def test_no_kernel_in_the_library_spills_registers(native):
    """every kernel of the shipped library keeps its working set in registers -- no spilled VGPRs and no
    private (scratch) memory -- read from the gfx950 code objects' metadata notes (tools/kernel_resources.py; needs the ROCm
    llvm tools, no GPU).  The one allowed exception is listed with its reason."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import kernel_resources as kr
    if not os.path.exists(os.path.join(kr.LLVM, "llvm-readelf")):
        pytest.skip("ROCm llvm tools not installed")
    ks = kr.kernels(native.LIB_PATH)
    assert len(ks) > 60                                        # every .hip file's kernels are in the bundle
    bad = [(k["name"], k["vgpr_spill"], k["scratch"]) for k in ks
           if (k["vgpr_spill"] > 0 or k["scratch"] > 0) and not any(a in k["name"] for a in kr.ALLOWED_SCRATCH)]      # (mangled names too)
    assert not bad, bad
    assert all(k["vgpr"] <= 512 for k in ks)                   # .vgpr_count = architectural + accumulation registers


def test_wgrad_tiling_picks_the_tn256_kernel_for_long_contractions():
    """Host logic (no launch): weight gradients of the encoders / heads at >= 4096 token rows go to the TN 256 kernel with
    tiles x splits in one round of the chip; short contractions and the 2 GB out_layer.fc1 matrix do not."""
    from lr2ppo_amd import ops
    for (M, N, K), want in (((768, 3072, 100864), (256, 7)), ((3072, 768, 100864), (256, 7)), ((2304, 768, 100864), (256, 9)),
                            ((768, 768, 100352), (256, 28)), ((768, 3072, 12544), (256, 7))):
        assert ops.choose_tiling(M, N, K, True, True) == want, (M, N, K)
    assert ops.choose_tiling(3072, 162816, 64, True, True)[0] != 256
    assert ops.choose_tiling(768, 3072, 2048, True, True)[0] != 256


def test_row_split_plan_fills_whole_rounds_and_leaves_a_thin_tail(native):
    """lr2_gemm_row_split_plan (round 4): rows of whole 256-tile rounds to the 256 x 256 kernel when the last round would be less
    than half full -- the shapes of the heads / RoBERTa at 2 tags -- and no split otherwise; ops.use_gemm256 asks for block_m = 256
    exactly where the plan (or a well-filled launch) exists.  Host arithmetic only."""
    import ctypes
    from lr2ppo_amd import ops
    lib = native.lib()

    def plan(M, N, K):
        r, t = ctypes.c_int(-1), ctypes.c_int(-1)
        assert lib.lr2_gemm_row_split_plan(M, N, K, ctypes.byref(r), ctypes.byref(t)) == 0
        return r.value, t.value
    assert plan(12544, 3072, 768) == (10752, 64)          # 588 tiles = 2.30 rounds: 42 tile rows (504 tiles) + 1792 rows
    assert plan(25088, 768, 3072) == (21760, 64)          # 294 tiles = 1.15 rounds: 85 tile rows (255 tiles) + 3328 rows
    assert plan(6272, 3072, 768) == (5376, 64)            # 300 tiles
    for shape in ((12544, 2304, 768), (100864, 3072, 768), (12544, 768, 3072), (12544, 768, 768), (25088, 3072, 768)):
        assert plan(*shape)[0] == 0, shape                # last round at least half full, or less than one round: one launch
    assert plan(8448, 2048, 96)[0] == 0                   # the tail kernels step K by 64
    assert lib.lr2_gemm_row_split_plan(0, 8, 8, None, None) != 0
    for M, N, K in ((12544, 3072, 768), (6272, 3072, 768), (25088, 768, 3072)):
        rows, _ = plan(M, N, K)
        tn = (N + 255) // 256
        assert rows % 256 == 0 and (rows // 256) * tn <= (((M + 255) // 256) * tn // 256) * 256
        assert ops.use_gemm256(M, N, K)
