"""Pin the CPU oracle to fixtures produced by importing the reference (oracle/gen_golden.py).

CPU-only; these are the `-m "not gpu"` checks that make the oracle trustworthy as the checker for
the HIP path.  Tolerances are fp32 round-off of re-ordered but algebraically identical ops.
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLD, load_golden
from oracle import lr2ppo_oracle as O


def _params(g, prefix):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


def test_key_specs_match_reference():
    with open(os.path.join(GOLD, "keys.json")) as f:
        keys = json.load(f)
    for kind in ("actor", "critic", "reward"):
        assert [(n, tuple(s)) for n, s in keys[kind]] == O.head_param_spec(kind)
    assert [(n, tuple(s)) for n, s in keys["vit_encoder"]] == O.encoder_param_spec(12, 768, 3072, True)
    assert [(n, tuple(s)) for n, s in keys["roberta_encoder"]] == O.encoder_param_spec(12, 768, 3072, False)
    assert [(n, tuple(s)) for n, s in keys["vit_embedding"]] == O.vit_embedding_spec(768, 3, 16, 197)
    assert [(n, tuple(s)) for n, s in keys["roberta_embedding"]] == O.text_embedding_spec(768, 50265, 514)
    n_actor = sum(int(np.prod(s)) for _, s in keys["actor"])
    n_critic = sum(int(np.prod(s)) for _, s in keys["critic"])
    assert (n_actor, n_critic) == (519070465, 526164481)


def test_xit_small_forward_and_grads():
    g = load_golden("xit_small.npz")
    P = {"xit." + k: v for k, v in _params(g, "param.").items()}
    out = O.xit(P, "xit", g["x"], g["y"])
    assert torch.allclose(out, g["out"], atol=2e-5, rtol=1e-5)
    out_self = O.xit(P, "xit", g["xs"], g["xs"])
    assert torch.allclose(out_self, g["out_self"], atol=2e-5, rtol=1e-5)
    assert int(g["causal_equals_full"]) == 1                           # quirk 2: causal mask is a no-op
    # gradients
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    x = g["x"].clone().requires_grad_(True)
    y = g["y"].clone().requires_grad_(True)
    (O.xit(Pg, "xit", x, y) * g["w"]).sum().backward()
    assert torch.allclose(x.grad, g["dx"], atol=1e-4, rtol=1e-4)
    assert torch.allclose(y.grad, g["dy"], atol=1e-4, rtol=1e-4)
    for k, v in Pg.items():
        ref = g["grad." + k[len("xit."):]]
        assert torch.allclose(v.grad, ref, atol=2e-4, rtol=1e-4), k


def test_losses():
    g = load_golden("losses.npz")
    assert abs(float(O.rank_loss(g["hand_scores"], g["hand_order"])) - 0.31) < 1e-6
    for tag, margin in (("hand", 0.01), ("zero", 0.01), ("rand", 0.01), ("rand5", 1.0)):
        got = O.rank_loss(g[tag + "_scores"], g[tag + "_order"], margin)
        assert torch.allclose(got, g[tag + "_rank"], atol=1e-7), tag
    assert float(g["zero_rank"]) == 0.0
    for clip, key in ((0.5, "vloss_05"), (0.2, "vloss_02")):
        assert torch.allclose(O.clipped_value_loss(g["v"], g["r"], g["ov"], clip), g[key], atol=1e-7)
    assert torch.allclose(O.masked_normalize(g["norm_in"]), g["norm_out"], atol=1e-6)


def test_adamw_matches_reference_update_order():
    g = load_golden("adamw.npz")
    wds = [0.01, 0.0, 0.01]
    for i in range(3):
        p = g[f"p0_{i}"]
        m = torch.zeros_like(p)
        v = torch.zeros_like(p)
        for step in range(3):
            p, m, v = O.adamw_step(p, g[f"g{step}_{i}"], m, v, lr=1e-2, wd=wds[i])
            assert torch.allclose(p, g[f"p{step + 1}_{i}"], atol=1e-7, rtol=1e-6)
            assert torch.allclose(m, g[f"m{step + 1}_{i}"], atol=1e-8, rtol=1e-6)
            assert torch.allclose(v, g[f"v{step + 1}_{i}"], atol=1e-10, rtol=1e-6)


def test_linear_schedule_table():
    with open(os.path.join(GOLD, "sched.json")) as f:
        s = json.load(f)
    for step, lr in enumerate(s["lrs"]):
        got = s["base_lr"] * O.linear_schedule_lambda(step, s["warmup_steps"], s["train_steps"])
        assert abs(got - lr) < 1e-12, step
    assert s["lrs"][0] == 0.0                                          # quirk 15: first cycle runs at lr 0


def test_ndcg_cases():
    g = load_golden("ndcg.npz")
    for i in range(int(g["n_cases"])):
        got = O.ndcg_vector(g[f"scores_{i}"], g[f"gold_{i}"])
        assert torch.allclose(got, g[f"ndcg_{i}"], atol=1e-6), i
    assert torch.all(g["ndcg_1"] == 1.0)                               # all-zero gold -> 1


def test_encoder_small_both_ln_placements():
    g = load_golden("encoder_small.npz")
    for tag in ("post", "pre"):
        P = _params(g, f"{tag}_param.")
        out = O.transformer_encoder(P, g[f"{tag}_emb"], g[f"{tag}_seg"], layers=2, heads=4, pre_ln=(tag == "pre"))
        assert torch.allclose(out, g[f"{tag}_out"], atol=5e-5, rtol=1e-4), tag
    assert torch.allclose(O.layernorm_tp(g["ln_x"], g["ln_gamma"], g["ln_beta"]), g["ln_out"], atol=1e-6)
    # the TP LayerNorm really differs from nn.LayerNorm (quirk 3)
    diff = (O.layernorm_torch(g["ln_x"], g["ln_gamma"], g["ln_beta"]) - g["ln_out"]).abs().max()
    assert diff > 1e-3


def test_embeddings_small():
    g = load_golden("embeddings_small.npz")
    out = O.vit_embedding(_params(g, "vit_param."), g["vit_img"], patch=8)
    assert torch.allclose(out, g["vit_out"], atol=2e-5, rtol=1e-5)
    out = O.text_embedding(_params(g, "txt_param."), g["txt_src"], g["txt_seg"])
    assert torch.allclose(out, g["txt_out"], atol=2e-5, rtol=1e-5)


def test_dropout_mask_statistics_and_determinism():
    k1 = O.dropout_keep_mask(seed=5, site=2, numel=200000, p=0.1)
    k2 = O.dropout_keep_mask(seed=5, site=2, numel=200000, p=0.1)
    k3 = O.dropout_keep_mask(seed=6, site=2, numel=200000, p=0.1)
    assert np.array_equal(k1, k2) and not np.array_equal(k1, k3)
    assert abs(k1.mean() - 0.9) < 5e-3
    # the two 16-bit fields of one hash (elements 2j, 2j + 1) and neighbouring hashes are uncorrelated; sites differ
    a = k1.astype(np.float64) - k1.mean()
    for lag in (1, 2, 3, 196, 197):
        assert abs((a[:-lag] * a[lag:]).mean() / a.var()) < 0.01, lag
    k4 = O.dropout_keep_mask(seed=5, site=3, numel=200000, p=0.1)
    assert abs(((k4.astype(np.float64) - k4.mean()) * a).mean() / a.var()) < 0.01
    for p in (0.05, 0.3, 0.5):
        assert abs(O.dropout_keep_mask(seed=11, site=0, numel=400000, p=p).mean() - (1 - p)) < 3e-3
    # attention masks: key dimension pitched to a multiple of 4
    m = O.attention_keep_mask(3, 1, 2, 3, 5, 0.1)
    flat = O.dropout_keep_mask(3, 1, 2 * 3 * 5 * 8, 0.1).reshape(2, 3, 5, 8)
    assert m.shape == (2, 3, 5, 5) and np.array_equal(m, flat[..., :5])


@pytest.mark.slow
def test_encoder_full_vit_and_roberta():
    g = load_golden("encoder_full.npz")
    with torch.no_grad():
        pe = O.seeded_params(O.vit_embedding_spec(768, 3, 16, 197), seed=61)
        pn = O.seeded_params(O.encoder_param_spec(12, 768, 3072, True), seed=62)
        gen = torch.Generator().manual_seed(63)
        img = torch.randn(2, 3, 224, 224, generator=gen)
        seg = torch.ones(2, 197, dtype=torch.long)
        e0 = O.vit_embedding(pe, img, 16)
        h = O.transformer_encoder(pn, e0, seg, 12, 12, True)
        assert torch.allclose(e0[:, :4], g["vit_emb_head"], atol=1e-4, rtol=1e-4)
        assert torch.allclose(h[:, :6], g["vit_hidden_head"], atol=2e-4, rtol=1e-3)
        assert torch.allclose(O.pooling_first(h, seg), g["vit_hidden_tok0"], atol=2e-4, rtol=1e-3)
        pe = O.seeded_params(O.text_embedding_spec(768, 50265, 514), seed=64)
        pn = O.seeded_params(O.encoder_param_spec(12, 768, 3072, False), seed=65)
        src, seg = g["txt_src"], g["txt_seg"]
        src2 = torch.randint(5, 50265, (2, 196), generator=gen)
        assert torch.equal(src, src2)                                # input stream reproducible from the seed
        e0 = O.text_embedding(pe, src, seg)
        h = O.transformer_encoder(pn, e0, seg, 12, 12, False)
        assert torch.allclose(e0[:, :4], g["txt_emb_head"], atol=1e-4, rtol=1e-4)
        assert torch.allclose(h[:, :6], g["txt_hidden_head"], atol=2e-4, rtol=1e-3)
        assert torch.allclose(h[:, -3:], g["txt_hidden_tail"], atol=2e-4, rtol=1e-3)


@pytest.mark.slow
def test_head_forward_full_size():
    g = load_golden("head_fwd.npz")
    bs, tags = int(g["bs"]), int(g["tags"])
    text, img, tgts = O.seeded_head_inputs(1234, bs, tags)
    state = torch.arange(tags).unsqueeze(0).repeat(bs, 1)
    with torch.no_grad():
        pa = O.seeded_params(O.head_param_spec("actor"), seed=7)
        loss, logits = O.actor_forward(pa, text, img, tgts)
        assert torch.allclose(logits, g["actor_logits"], atol=2e-5, rtol=1e-4)
        assert torch.allclose(loss, g["actor_loss"], atol=1e-5)
        text5, img5, tg5 = O.seeded_head_inputs(4321, 1, 5)
        _, lg5 = O.actor_forward(pa, text5, img5, tg5)
        assert torch.allclose(lg5, g["actor_logits_eval5"], atol=2e-5, rtol=1e-4)
        del pa
        pc = O.seeded_params(O.head_param_spec("critic"), seed=8)
        v = O.critic_forward(pc, text, img, state)
        assert torch.allclose(v, g["critic_value"], atol=2e-5, rtol=1e-4)
        vf = O.critic_forward(pc, text, img, state.flip(dims=[-1]))
        assert torch.allclose(vf, g["critic_value_flipped"], atol=2e-5, rtol=1e-4)
        del pc
        nxt = O.rollout_next_state(logits.view(bs, tags), state)
        assert torch.equal(nxt, g["next_state"])
        pr = O.seeded_params(O.head_param_spec("reward"), seed=9)
        r = O.reward_forward(pr, text, img, nxt)
        assert torch.allclose(r, g["reward"], atol=2e-5, rtol=1e-4)


def _check_train_steps(gold_name, kind, seed, make_batch, loss_fn, n_out, steps_checked=2):
    """The first `steps_checked` of the fixture's steps (step 0 runs at lr 0, step 1 is the first real AdamW update): a full-size
    head step costs the CPU oracle ~1 minute; the fixture's later steps are checked on the HIP path (tests/test_stages_gpu.py)."""
    g = load_golden(gold_name)
    bs, tags, steps = int(g["bs"]), int(g["tags"]), min(int(g["steps"]), steps_checked)
    P = O.seeded_params(O.head_param_spec(kind), seed=seed)
    batches = [make_batch(g, step, bs, tags) for step in range(steps)]
    for step in range(steps):
        assert abs(float(g[f"lr_{step}"]) - 1e-3 * O.linear_schedule_lambda(step, 2.1, 21)) < 1e-12
    outs = O.sgd_free_train_steps(P, loss_fn, batches, 1e-3, 2.1, 21)
    for step in range(steps):
        ref = float(g[f"loss_{step}"])
        assert abs(float(outs[step][0]) - ref) < 2e-5 * max(1.0, abs(ref)), step
        if n_out > 1:
            assert float(outs[step][1]) == float(g[f"acc_{step}"]), step
    last = steps - 1
    for key in [k for k in g if k.startswith(f"w{last}.")]:
        n = key[len(f"w{last}."):]
        got = P[n].flatten()[g["idx." + n]]
        assert (got - g[key]).abs().max() < 2e-6, n
    return g, P


@pytest.mark.slow
def test_stage1_pointwise_train_steps_match_reference():
    """pointwise.train_model steps (Actor architecture + SmoothL1 + per-batch scheduler): losses, lr and the
    sampled weights after the first real update.  (The fixture's third step and its evaluation logits: GPU suite.)"""
    _check_train_steps("stage1_step.npz", "actor", 17,
                       lambda g, step, bs, tags: O.seeded_head_inputs(2000 + step, bs, tags), O.stage1_loss, 1)


@pytest.mark.slow
def test_stage2_pair_reward_train_steps_match_reference():
    """reward_pair_dataloader.train_model steps (two forwards + hinge + AdamW)."""
    def batch(g, step, bs, tags):
        text, img, _ = O.seeded_head_inputs(3000 + step, bs, tags)
        return text, img, g[f"chosen_{step}"], g[f"reject_{step}"]
    _check_train_steps("stage2_step.npz", "reward", 19, batch, O.stage2_loss, 2)


def test_stage2_get_index_restatement():
    import json
    with open(os.path.join(GOLD, "stage2_get_index.json")) as f:
        cases = json.load(f)
    assert len(cases) == 5
    for c in cases:
        ch, rj = O.pair_index(c["targets"], c["order"])
        assert ch == c["chosen"] and rj == c["reject"], c


def _enc_bwd_case(tag):
    spec = O.encoder_param_spec(2, 128, 256, tag == "pre")
    P = O.seeded_params(spec, seed=51, std=0.15, skip_gamma_beta=False)
    g = torch.Generator().manual_seed(52)
    emb = torch.randn(3, 50, 128, generator=g)
    wout = torch.randn(3, 50, 128, generator=g)
    seg = torch.ones(3, 50, dtype=torch.long)
    seg[1, 33:] = 0
    seg[2, 7:] = 0
    return P, emb, wout, seg


@pytest.mark.parametrize("tag", ["post", "pre"])
def test_encoder_backward_oracle_matches_reference_autograd(tag):
    """A14 backward: the oracle's encoder under torch autograd against gradients frozen from the reference's own modules."""
    g = load_golden("encoder_bwd_small.npz")
    P, emb, wout, seg = _enc_bwd_case(tag)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    e = emb.clone().requires_grad_(True)
    out = O.transformer_encoder(Pg, e, seg, 2, 2, tag == "pre")
    (out * wout).sum().backward()
    assert (out - g[f"{tag}_out"]).abs().max() < 2e-5
    assert (e.grad - g[f"{tag}_demb"]).abs().max() < 2e-5 * max(1.0, float(g[f"{tag}_demb"].abs().max()))
    for k in P:
        ref = g[f"{tag}_grad.{k}"]
        assert (Pg[k].grad - ref).abs().max() < 2e-5 * max(1.0, float(ref.abs().max())), k


def test_embedding_backward_oracle_matches_reference_autograd():
    """A15 backward: the oracle's two embedding compositions under autograd against the reference modules' gradients."""
    g = load_golden("embeddings_bwd_small.npz")
    P = {k: v.clone().requires_grad_(True) for k, v in O.seeded_params(O.vit_embedding_spec(32, 3, 8, 25), seed=51, std=0.5).items()}
    out = O.vit_embedding(P, g["vit_img"], 8)
    (out * g["vit_w"]).sum().backward()
    assert (out - g["vit_out"]).abs().max() < 1e-4
    for k in P:
        ref = g["vit_grad." + k]
        assert (P[k].grad - ref).abs().max() < 1e-4 * max(1.0, float(ref.abs().max())), k
    P = {k: v.clone().requires_grad_(True)
         for k, v in O.seeded_params(O.text_embedding_spec(32, 100, 20), seed=53, std=0.5, skip_gamma_beta=False).items()}
    out = O.text_embedding(P, g["txt_src"], g["txt_seg"])
    (out * g["txt_w"]).sum().backward()
    assert (out - g["txt_out"]).abs().max() < 1e-5
    for k in P:
        ref = g["txt_grad." + k]
        assert (P[k].grad - ref).abs().max() < 1e-5 * max(1.0, float(ref.abs().max())), k


def test_trad_classifier_train_steps_match_reference():
    """BASELINE configs[0]: the seq-len-1 `_trad` head (finetune/pointwise_trad.py) -- three train steps + inference."""
    g = load_golden("trad_step.npz")
    steps = int(g["steps"])
    P = O.seeded_params(O.trad_param_spec(), seed=27)
    batches = [(g[f"feats_{s}"], g[f"tgts_{s}"]) for s in range(steps)]
    outs = O.sgd_free_train_steps(P, lambda Pg, b: O.trad_forward(Pg, b[0], b[1]), batches, 1e-3, 2.1, 21)
    for s in range(steps):
        ref = float(g[f"loss_{s}"])
        assert abs(float(outs[s][0]) - ref) < 2e-5 * max(1.0, abs(ref)), s
    for key in [k for k in g if k.startswith(f"w{steps - 1}.")]:
        n = key[len(f"w{steps - 1}."):]
        assert (P[n].flatten()[g["idx." + n]] - g[key]).abs().max() < 2e-6, n
    with torch.no_grad():
        logits = O.trad_forward(P, g["feats_0"])
    ref = g["eval_logits"]
    assert (logits - ref).abs().max() < 1e-4 * max(1.0, float(ref.abs().max()))


# ---- round 2 fixtures: encoder backward at ViT-B/16 / RoBERTa-base width, DualEmbedding + DualEncoder ----------------------
def sq_close(got, ref, rel):
    """Sums of squares agree to `rel`; a gradient that is analytically zero (the key bias: softmax is shift-invariant
    along the keys) is rounding noise on both sides and only has to stay tiny."""
    return got < 1e-9 if ref < 1e-9 else abs(got / ref - 1.0) < rel


def enc_bwd_wide_case(tag):
    """Inputs of oracle/gen_golden.py::gen_encoder_bwd_wide, rebuilt from the same seeds (shared with the GPU test)."""
    L = 197 if tag == "pre" else 196
    P = O.seeded_params(O.encoder_param_spec(2, 768, 3072, tag == "pre"), seed=71, std=0.05, skip_gamma_beta=False)
    g = torch.Generator().manual_seed(72)
    emb = torch.randn(2, L, 768, generator=g)
    wout = torch.randn(2, L, 768, generator=g)
    seg = torch.ones(2, L, dtype=torch.long)
    if tag == "post":
        seg[1, 131:] = 0
    return P, emb, wout, seg


@pytest.mark.slow
@pytest.mark.parametrize("tag", ["post", "pre"])
def test_encoder_backward_wide_oracle_matches_reference_autograd(tag):
    """A14 backward at hidden 768 / 12 heads / L = 197 and 196-with-padding: oracle autograd vs the reference's (sampled)."""
    g = load_golden("encoder_bwd_wide.npz")
    P, emb, wout, seg = enc_bwd_wide_case(tag)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    e = emb.clone().requires_grad_(True)
    out = O.transformer_encoder(Pg, e, seg, 2, 12, tag == "pre")
    (out * wout).sum().backward()
    idx = g[f"{tag}_idx"]
    assert (out.detach().flatten()[idx] - g[f"{tag}_out"]).abs().max() < 5e-5
    ref = g[f"{tag}_demb"]
    assert (e.grad.flatten()[idx] - ref).abs().max() < 5e-5 * max(1.0, float(ref.abs().max()))
    assert abs(float((e.grad.double() ** 2).sum()) / float(g[f"{tag}_demb_sq"]) - 1.0) < 1e-4
    for k in P:
        ref = g[f"{tag}_grad.{k}"]
        got = Pg[k].grad.flatten()[g[f"{tag}_gidx.{k}"]]
        assert (got - ref).abs().max() < 5e-5 * max(1.0, float(ref.abs().max())), k
        assert sq_close(float((Pg[k].grad.double() ** 2).sum()), float(g[f"{tag}_gsq.{k}"]), 1e-4), k


def dual_case(tag):
    """(embedding params, encoder params, kinds, pre_ln flags, tied) of oracle/gen_golden.py::gen_dual."""
    import json
    keys = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "dual_keys.json")))[tag]
    pe = O.seeded_params([(n, tuple(s)) for n, s in keys["embedding"]], seed=81, std=0.3, skip_gamma_beta=False)
    pn = O.seeded_params([(n, tuple(s)) for n, s in keys["encoder"]], seed=82, std=0.15, skip_gamma_beta=False)
    kinds = ("text", "vit") if tag == "tv" else ("text", "text")
    pre_ln = (False, True) if tag == "tv" else (False, False)
    return pe, pn, kinds, pre_ln, tag == "tt"


@pytest.mark.parametrize("tag", ["tv", "tt"])
def test_dual_embedding_and_encoder_oracle_match_reference(tag):
    """A15: DualEmbedding (inner + stream LayerNorms) and DualEncoder (incl. tie_weights), forward and parameter gradients."""
    g = load_golden("dual.npz")
    pe, pn, kinds, pre_ln, tied = dual_case(tag)
    Pe = {k: v.clone().requires_grad_(True) for k, v in pe.items()}
    Pn = {k: v.clone().requires_grad_(True) for k, v in pn.items()}
    src, seg = (g[f"{tag}_src0"], g[f"{tag}_src1"]), (g[f"{tag}_seg0"], g[f"{tag}_seg1"])
    e = O.dual_embedding(Pe, src, seg, kinds, patch=8, tied=tied)
    h = O.dual_encoder(Pn, e, seg, 1, 2, pre_ln, tied=tied)
    ((h[0] * g[f"{tag}_w0"]).sum() + (h[1] * g[f"{tag}_w1"]).sum()).backward()
    for i in range(2):
        assert (e[i] - g[f"{tag}_e{i}"]).abs().max() < 2e-5
        assert (h[i] - g[f"{tag}_h{i}"]).abs().max() < 5e-5
    for k, v in Pe.items():
        ref = g[f"{tag}_egrad.{k}"]
        assert (v.grad - ref).abs().max() < 5e-5 * max(1.0, float(ref.abs().max())), k
    for k, v in Pn.items():
        ref = g[f"{tag}_ngrad.{k}"]
        assert (v.grad - ref).abs().max() < 5e-5 * max(1.0, float(ref.abs().max())), k


@pytest.mark.slow
def test_cls_mode_oracle_matches_reference():
    """mode = 'cls': 3-way actor head, NLL loss, expected-label scores (rollout: softmax; evaluate: raw logits), rollout glue
    and the update's loss / gradients against the reference's own run (tests/golden/cls_step.npz)."""
    g = load_golden("cls_step.npz")
    bs, tags = int(g["bs"]), int(g["tags"])
    Pa = O.seeded_params(O.head_param_spec("actor", n_out=3), seed=17)
    text, img, tgts = O.seeded_head_inputs(2000, bs, tags)
    state = torch.arange(tags).unsqueeze(0).repeat(bs, 1)
    with torch.no_grad():
        loss, logits = O.actor_forward_cls(Pa, text, img, tgts)
        scores = O.cls_action_scores(logits, bs, tags)
        nxt = O.rollout_next_state(scores, state)
    # (critic / reward are the 'reg'-mode functions, pinned by head_fwd.npz / train_step.npz; their fixture values are used)
    assert (logits - g["logits"]).abs().max() < 2e-5 and abs(float(loss) - float(g["nll"])) < 1e-5
    assert (scores - g["scores"]).abs().max() < 1e-5 and torch.equal(nxt, g["next_state"])
    assert (O.cls_action_scores(logits, bs * tags, 1, softmax=False).view(-1) - g["eval_scores"]).abs().max() < 2e-5
    Pg = {k: v.clone().requires_grad_(True) for k, v in Pa.items()}
    new_logits = O.actor_forward_cls(Pg, text, img, None)
    new_scores = O.cls_action_scores(new_logits, bs, tags)
    new_value = g["value"]              # the critic is unchanged between the rollout and the update's forward (eval mode)
    pl, vl, ex = O.ppo_update_math(new_scores, new_value, g["scores"], g["reward"], g["value"], g["next_state"], 0.001, 0.001, 0.5)
    pl.backward()
    m = g["metrics"]
    assert abs(float(pl) - float(m[0])) < 1e-5 and abs(float(vl) - float(m[1])) < 1e-5 and abs(float(ex["rank_loss"]) - float(m[8])) < 1e-5
    for key in [k for k in g if k.startswith("g.actor.")]:
        n = key[len("g.actor."):]
        ref = g[key]
        got = Pg[n].grad.flatten()[g["idx.actor." + n]]
        assert (got - ref).abs().max() < 1e-6 + 2e-4 * float(ref.abs().max()), n


def test_ppo_trad_oracle_matches_reference():
    """finetune/ppo_trad.py's Actor / Critic / Reward at sequence length 1 (trad_actor_forward, trad_critic_forward) against
    the imported reference's rollout tensors at the initial weights (ppo_trad_step.npz, cycle 0)."""
    g = load_golden("ppo_trad_step.npz")
    bs, tags = int(g["bs"]), int(g["tags"])
    Pa = O.seeded_params(O.trad_head_param_spec("actor"), seed=37)
    Pc = O.seeded_params(O.trad_head_param_spec("critic"), seed=38)
    Pr = O.seeded_params(O.trad_head_param_spec("reward"), seed=39)
    state = torch.arange(tags).unsqueeze(0).repeat(bs, 1)
    for mb in range(2):
        k = f"c0_mb{mb}_"
        with torch.no_grad():
            scores = O.trad_actor_forward(Pa, g[k + "text"]).view(bs, tags)
            value = O.trad_critic_forward(Pc, g[k + "text"], state)
            nxt = O.rollout_next_state(scores, state)
            r = O.trad_critic_forward(Pr, g[k + "text"], nxt, n_pos=4)
        assert (scores - g[k + "scores"]).abs().max() < 2e-5 and (value - g[k + "value"]).abs().max() < 2e-5
        assert torch.equal(nxt, g[k + "next_state"]) and (r - g[k + "reward"]).abs().max() < 2e-5
